#!/bin/bash
# same-box A/B of the product library before / after the knob cleanup, both GEMM modes; then the GPU suite
OUT=/root/repo/gpurun_out/${1:-r3clean}; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for g in fp32 bf16x3; do for c in cfg2 cfg4 cfg3 cfg5; do
  echo "== $c $g" | tee -a $OUT/ab.txt
  PYDENS_AMD_GEMM=$g timeout 300 python tools/kbench.py $c gpurun_variants/lib_pre_cleanup.so pydens_amd/libpinn_hip.so gpurun_variants/lib_pre_cleanup.so pydens_amd/libpinn_hip.so 2>&1 | grep tile | tee -a $OUT/ab.txt
done; done
bash tools/gpu_tests.sh ${1:-r3clean}
