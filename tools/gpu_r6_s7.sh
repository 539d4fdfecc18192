#!/bin/bash
# round 6 session 7b: the residual-program kernel as two independent workgroups per CU (16-point tiles, W^T from the global copy)
TAG=${1:-r6s7}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 600 python tools/kbench.py program $V/lib_p_base.so $V/lib_p_w2.so > $OUT/kbench_program.txt 2>&1; tail -n 4 $OUT/kbench_program.txt
