#!/bin/bash
# A/B of the chain kernel against the tile kernels on one box: tools/gpu_chain.sh <tag> [libs...]
TAG=${1:-chain}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
LIBS=${@:-gpurun_variants/lib_chain.so gpurun_variants/lib_nochain.so}
for c in cfg2 cfg4; do
  timeout 300 python tools/kbench.py $c $LIBS > $OUT/kbench_$c.txt 2>&1
done
cat $OUT/kbench_cfg2.txt $OUT/kbench_cfg4.txt
