#!/bin/bash
# same-box A/B of the fp32 kernels: round-2 end state (gpurun_variants/r2tree = git worktree of e23ba81, built in place) vs this tree
OUT=/root/repo/gpurun_out/${1:-r2ab}; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for rep in 1 2; do for c in cfg2 cfg4 cfg3 cfg5; do
  (cd gpurun_variants/r2tree && timeout 300 python tools/kbench.py $c pydens_amd/libpinn_hip.so 2>&1 | grep tile | sed 's/^/r2   /') | tee -a $OUT/ab.txt
  timeout 300 python tools/kbench.py $c pydens_amd/libpinn_hip.so 2>&1 | grep tile | sed 's/^/now  /' | tee -a $OUT/ab.txt
done; done
