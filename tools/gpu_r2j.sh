#!/bin/bash
OUT=/root/repo/gpurun_out/r2j; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 300 python tools/kbench.py cfg2 $V/lib_base.so $V/lib_c2w2.so $V/lib_c2w2ns.so > $OUT/kb_cfg2.txt 2>&1; cat $OUT/kb_cfg2.txt
