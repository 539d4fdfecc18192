#!/bin/bash
# round 6 session 4: same-box A/B of the micro-structure knobs (point-stage wave per team, fast IC gate, early LDS read of the bias rows) on configs 2 / 4
TAG=${1:-r6s4}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
LIBS="$V/lib_r6b_base.so $V/lib_r6b_k1.so $V/lib_r6b_k2.so $V/lib_r6b_k3.so $V/lib_r6b_all.so"
timeout 600 python tools/kbench.py cfg4 $LIBS > $OUT/kbench_cfg4.txt 2>&1; tail -10 $OUT/kbench_cfg4.txt
timeout 600 python tools/kbench.py cfg2 $LIBS > $OUT/kbench_cfg2.txt 2>&1; tail -10 $OUT/kbench_cfg2.txt
