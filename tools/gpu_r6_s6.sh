#!/bin/bash
# round 6 session 6: fit rates after the lean fp64 sincos / DPP fp64 sums, kernel time of config 2 (product vs the build before), the GPU suite
TAG=${1:-r6s6}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 400 python tools/small_fit_rate.py > $OUT/small_fit_rate.txt 2>&1; head -8 $OUT/small_fit_rate.txt
timeout 600 python tools/kbench.py cfg2 $V/lib_r6_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_cfg2.txt 2>&1; tail -4 $OUT/kbench_cfg2.txt
timeout 600 python tools/kbench.py cfg4 $V/lib_r6_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_cfg4.txt 2>&1; tail -4 $OUT/kbench_cfg4.txt
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -8 $OUT/pytest_gpu.txt
cp gpurun_out/grad_margins.txt $OUT/ 2>/dev/null
