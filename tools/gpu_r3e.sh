#!/bin/bash
OUT=/root/repo/gpurun_out/r3e; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 300 python tools/split_check.py $V/lib_sp_a.so > $OUT/check.txt 2>&1; grep cfg $OUT/check.txt
for c in cfg2 cfg4; do
  timeout 200 python tools/kbench.py $c $V/lib_sp_a.so 2>&1 | grep tile | tail -1
  PYDENS_AMD_GEMM=bf16x3 timeout 400 python tools/kbench.py $c $V/lib_sp_a.so $V/lib_sp_b.so $V/lib_sp_c.so 2>&1 | grep tile | tail -3
done
for c in cfg2 cfg4; do
  PYDENS_AMD_GEMM=bf16x3 timeout 200 python tools/phases.py $V/lib_sp_ph.so $c > $OUT/ph_${c}_split.txt 2>&1
  grep -v Warn $OUT/ph_${c}_split.txt | cut -c1-70
done
