#!/bin/bash
OUT=/root/repo/gpurun_out/r3t; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for c in cfg5 cfg3; do
  timeout 200 python tools/phases.py gpurun_variants/lib_wph.so $c > $OUT/ph_${c}_fp32.txt 2>&1
  PYDENS_AMD_GEMM=bf16x3 timeout 200 python tools/phases.py gpurun_variants/lib_wph.so $c > $OUT/ph_${c}_split.txt 2>&1
  paste $OUT/ph_${c}_fp32.txt $OUT/ph_${c}_split.txt | cut -c1-60,95-160 | grep -v Warn
done
