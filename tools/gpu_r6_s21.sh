#!/bin/bash
# round 6 session 21: experiment -- the WGX tile kernels of BASELINE configs 3 / 5 with tanh fixed at compile time (lib_tanhst.so) against the one-bit run-time activation code (product)
TAG=${1:-r6s21}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 900 python tools/kbench.py cfg3 pydens_amd/libpinn_hip.so $V/lib_tanhst.so > $OUT/kbench_cfg3.txt 2>&1; tail -n 4 $OUT/kbench_cfg3.txt
timeout 900 python tools/kbench.py cfg5 pydens_amd/libpinn_hip.so $V/lib_tanhst.so > $OUT/kbench_cfg5.txt 2>&1; tail -n 4 $OUT/kbench_cfg5.txt
