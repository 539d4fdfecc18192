#!/bin/bash
# A/B of team-local barriers (PINN_TEAM_FLAGS): tools/gpu_tf.sh <tag> lib1 lib2 ...  (kbench cfg2 + cfg4, both GEMM modes)
TAG=$1; shift; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for g in fp32 bf16x3; do for c in cfg2 cfg4; do
  echo "== $c $g" | tee -a $OUT/kb.txt
  PYDENS_AMD_GEMM=$g timeout 300 python tools/kbench.py $c "$@" 2>&1 | grep tile | tee -a $OUT/kb.txt | tail -n $#
done; done
