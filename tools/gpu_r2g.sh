#!/bin/bash
OUT=/root/repo/gpurun_out/r2g; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 300 python tools/kbench.py cfg4 $V/lib_base.so $V/lib_c4regb.so $V/lib_c4regbns.so $V/lib_c4ns.so > $OUT/kb_cfg4.txt 2>&1; cat $OUT/kb_cfg4.txt
timeout 200 python tools/fit_one.py cfg1 3000 > $OUT/fit_cfg1.txt 2>&1; tail -1 $OUT/fit_cfg1.txt
