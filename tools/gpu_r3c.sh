#!/bin/bash
OUT=/root/repo/gpurun_out/r3c; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
for c in cfg2 cfg4; do
  timeout 200 python tools/phases.py $V/lib_sp_ph.so $c > $OUT/ph_${c}_fp32.txt 2>&1
  PYDENS_AMD_GEMM=bf16x3 timeout 200 python tools/phases.py $V/lib_sp_ph.so $c > $OUT/ph_${c}_split.txt 2>&1
  paste $OUT/ph_${c}_fp32.txt $OUT/ph_${c}_split.txt | cut -c1-250 | grep -v Warn
done
