#!/bin/bash
# A/B of variant builds on the wide configs (bf16x3 mode): tools/gpu_abw.sh <tag> lib1 lib2 ...
TAG=$1; shift; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for c in cfg3 cfg5; do
  PYDENS_AMD_GEMM=bf16x3 timeout 600 python tools/kbench.py $c "$@" 2>&1 | grep tile | tee -a $OUT/kb_$c.txt | tail -n $#
done
