#!/bin/bash
# split kernels without the SLP vectoriser: timing (two teams / two workgroups per CU) and repeatability. tools/gpu_var2i.sh [tag]
TAG=${1:-var2i}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
for c in cfg2 cfg4; do
  echo "== $c bf16x3" | tee -a $OUT/kb.txt
  PYDENS_AMD_GEMM=bf16x3 timeout 400 python tools/kbench.py $c $V/lib_base.so $V/lib_base_noslp.so $V/lib_base_noslp11.so $V/lib_v2_noslp.so $V/lib_v2_noslp11.so 2>&1 | grep tile | tee -a $OUT/kb.txt
done
timeout 600 python tools/repeat_check.py cfg2,cfg4 $V/lib_v2_noslp.so,$V/lib_v2_noslp11.so,$V/lib_base_noslp.so --caps 0 --gemms bf16x3 --reps 30 > $OUT/repeat.txt 2>&1
grep distinct $OUT/repeat.txt
