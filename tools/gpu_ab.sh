#!/bin/bash
# A/B of split-kernel variant builds: tools/gpu_ab.sh <tag> lib1 lib2 ...   (kbench cfg2 + cfg4, bf16x3 mode)
TAG=$1; shift; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for c in cfg2 cfg4; do
  PYDENS_AMD_GEMM=bf16x3 timeout 600 python tools/kbench.py $c "$@" 2>&1 | grep tile | tee -a $OUT/kb_$c.txt | tail -n $#
done
