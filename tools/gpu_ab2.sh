#!/bin/bash
# same-box A/B of experiment builds on the split kernels of configs 2 / 4 (kbench, both interleaved twice) + a repeat check:
#   tools/gpu_ab2.sh <tag> lib1 lib2 ...
TAG=$1; shift; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for c in cfg2 cfg4; do
  echo "== $c bf16x3" | tee -a $OUT/kb.txt
  PYDENS_AMD_GEMM=bf16x3 timeout 400 python tools/kbench.py $c "$@" 2>&1 | grep tile | tee -a $OUT/kb.txt
done
LIBS=$(echo "$@" | tr ' ' ',')
timeout 600 python tools/repeat_check.py cfg2,cfg4 $LIBS --caps 0 --gemms bf16x3 --reps 6 > $OUT/repeat.txt 2>&1
grep distinct $OUT/repeat.txt
