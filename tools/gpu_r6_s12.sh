#!/bin/bash
# round 6 session 12: the generic / breadth kernels after the direction-code extensions went behind the stream shape (against the build before u_xyz), the three-column GPU test
TAG=${1:-r6s12}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 900 python tools/kbench.py gelu256 $V/lib_p_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_gelu256.txt 2>&1; tail -n 4 $OUT/kbench_gelu256.txt
timeout 600 python tools/kbench.py sin128 $V/lib_p_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_sin128.txt 2>&1; tail -n 4 $OUT/kbench_sin128.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "third_order_direction_groups or golden_extra or fourth_order" > $OUT/pytest_dirs.txt 2>&1; tail -n 3 $OUT/pytest_dirs.txt
