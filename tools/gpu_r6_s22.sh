#!/bin/bash
# round 6 session 22: Tanh fixed at compile time in the light-skip kernels (skip128 / skip256) against the build before; the wide BASELINE / skip tests on the final library
TAG=${1:-r6s22}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 600 python tools/kbench.py skip128 $V/lib_p_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_skip128.txt 2>&1; tail -n 4 $OUT/kbench_skip128.txt
timeout 900 python tools/kbench.py skip256 $V/lib_p_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_skip256.txt 2>&1; tail -n 4 $OUT/kbench_skip256.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_occupancy.py tests/test_split_gemm.py -m gpu -q -k "cfg3 or cfg5 or skip or split or soak or bitwise or golden or bench_parity or w100" > $OUT/pytest_wide.txt 2>&1; tail -n 3 $OUT/pytest_wide.txt
