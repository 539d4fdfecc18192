#!/bin/bash
# round 5, session 12: the product build with BASELINE config 2's kernel in a unit of its own (default scheduler) against the iterative-ilp build
TAG=${1:-r5s}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
L="gpurun_variants/lib_r5x_ilp.so pydens_amd/libpinn_hip.so gpurun_variants/lib_r5x_dflt.so"
timeout 400 python tools/kbench.py cfg2 $L > $OUT/kbench_cfg2.txt 2>&1; grep -v amdgpu $OUT/kbench_cfg2.txt | tail -6
timeout 400 python tools/kbench.py cfg4 $L > $OUT/kbench_cfg4.txt 2>&1; grep -v amdgpu $OUT/kbench_cfg4.txt | tail -6
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "golden and cfg2" > $OUT/pytest_cfg2.log 2>&1; tail -2 $OUT/pytest_cfg2.log
