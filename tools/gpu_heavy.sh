#!/bin/bash
# full breadth kernels with streamed weight gradients (round 4): parity, then the A/B against the read-modify-write build
TAG=${1:-heavy}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_occupancy.py tests/test_fuzz_equations.py -m gpu -q -x --durations=8 \
  -k "wide_residual or skip128 or layout or wide or third_order or every_width" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -14 $OUT/pytest.log
bash tools/gpu_ab_any.sh $TAG "sin128 gelu256" gpurun_variants/lib_heavy_rmw.so
