#!/bin/bash
# round 5, session 15: scheduler strategy of the width-16 unit (BASELINE config 1: one-CU fit chunk and launch graphs), same box
TAG=${1:-r5v}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for pass in 1 2; do for l in product n16_maxilp; do for m in 2 0; do
  if [ $l = product ]; then unset SMALL_FIT_LIB; else export SMALL_FIT_LIB=/root/repo/gpurun_variants/lib_$l.so; fi
  PYDENS_AMD_FIT_PERSIST=$m timeout 200 python tools/small_fit_rate.py cfg1 12800 2>&1 | grep "^cfg1" | sed "s/^/$l mode $m pass $pass: /"
done; done; done | tee $OUT/n16.txt
