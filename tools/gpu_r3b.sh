#!/bin/bash
# round 3: first device run of the split-bf16 kernels (variant builds): agreement with the exact kernels, kernel times
OUT=/root/repo/gpurun_out/r3b; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 300 python tools/split_check.py $V/lib_sp_noslp.so > $OUT/check.txt 2>&1; tail -5 $OUT/check.txt
for c in cfg2 cfg4; do
  timeout 300 python tools/kbench.py $c $V/lib_sp_noslp.so > $OUT/kb_${c}_fp32.txt 2>&1
  PYDENS_AMD_GEMM=bf16x3 timeout 400 python tools/kbench.py $c $V/lib_sp_noslp.so $V/lib_sp_slp.so $V/lib_sp_noslp_ilp.so > $OUT/kb_${c}_split.txt 2>&1
  grep tile $OUT/kb_${c}_fp32.txt $OUT/kb_${c}_split.txt
done
