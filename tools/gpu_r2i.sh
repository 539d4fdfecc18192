#!/bin/bash
OUT=/root/repo/gpurun_out/r2i; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
for f in 0 8 16 32 24 56; do echo "flags $f"; timeout 300 python tools/kbench.py --flags=$f cfg5 $V/lib_base.so 2>&1 | grep -v amdgpu | head -1; done > $OUT/kb_cfg5_flags.txt 2>&1; cat $OUT/kb_cfg5_flags.txt
