#!/bin/bash
OUT=/root/repo/gpurun_out/r2p; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 300 python tools/kbench.py cfg4 $V/lib_base.so $V/lib_skew.so > $OUT/kb_cfg4.txt 2>&1; cat $OUT/kb_cfg4.txt
timeout 300 python tools/kbench.py cfg2 $V/lib_base.so $V/lib_skew.so > $OUT/kb_cfg2.txt 2>&1; cat $OUT/kb_cfg2.txt
