#!/bin/bash
# round 6 session 5: unrounded pre-pass constants + activation parameters on the device; A/B of PTALL on config 4 with the fast gate, the
# top-barrier knob, the LDS-resident slab of config 2; small-batch fit rates (the fp64 pre-pass in the one-CU chunk); the new GPU tests
TAG=${1:-r6s5}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 600 python tools/cfg4_bl_probe.py 99 100 101 > $OUT/cfg4_bl_probe.txt 2>&1; grep -A3 "^--- fp32" $OUT/cfg4_bl_probe.txt | head -14
timeout 600 python tools/kbench.py cfg4 $V/lib_r6c_base.so $V/lib_r6c_ptall2.so $V/lib_r6c_nt.so > $OUT/kbench_cfg4.txt 2>&1; tail -6 $OUT/kbench_cfg4.txt
timeout 600 python tools/kbench.py cfg2 $V/lib_r6c_base.so $V/lib_r6c_nt.so $V/lib_r6c_slabl.so pydens_amd/libpinn_hip.so > $OUT/kbench_cfg2.txt 2>&1; tail -8 $OUT/kbench_cfg2.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bench_parity_rule or breadth_features or golden or layout_breadth" > $OUT/pytest_subset.txt 2>&1; tail -4 $OUT/pytest_subset.txt
timeout 400 python tools/small_fit_rate.py > $OUT/small_fit_rate.txt 2>&1; head -8 $OUT/small_fit_rate.txt
