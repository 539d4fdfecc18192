#!/bin/bash
OUT=/root/repo/gpurun_out/r2l; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 300 python tools/kbench.py cfg5 $V/lib_base.so $V/lib_s256def.so $V/lib_s256ilp.so > $OUT/kb_cfg5.txt 2>&1; cat $OUT/kb_cfg5.txt
timeout 300 python tools/kbench.py cfg3 $V/lib_base.so $V/lib_s128mo.so $V/lib_s128ilp.so > $OUT/kb_cfg3.txt 2>&1; cat $OUT/kb_cfg3.txt
timeout 300 python tools/kbench.py cfg2 $V/lib_base.so $V/lib_s64mo.so $V/lib_s64ilp.so $V/lib_teamns.so > $OUT/kb_cfg2.txt 2>&1; cat $OUT/kb_cfg2.txt
timeout 300 python tools/kbench.py cfg4 $V/lib_base.so $V/lib_s64mo.so $V/lib_s64ilp.so $V/lib_teamns.so > $OUT/kb_cfg4.txt 2>&1; cat $OUT/kb_cfg4.txt
