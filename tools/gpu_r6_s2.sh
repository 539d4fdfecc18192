#!/bin/bash
# round 6 session 2: the fp64 pre-pass on the device -- the bench parity rule as a test (30 cases), the config-4 probe again, the default
# bench line (with baseline_configs + sustained; wall time of the whole command), kernel time against the build before the change
TAG=${1:-r6s2}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "bench_parity_rule" > $OUT/pytest_parity_rule.txt 2>&1; tail -5 $OUT/pytest_parity_rule.txt
timeout 600 python tools/cfg4_bl_probe.py 99 100 101 > $OUT/cfg4_bl_probe.txt 2>&1; grep -A3 "^---" $OUT/cfg4_bl_probe.txt | head -40
( time timeout 900 python bench.py > $OUT/bench_default.txt 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "bench rc=$?"; cat $OUT/bench_default.time; tail -c 6000 $OUT/bench_default.txt
timeout 600 python tools/kbench.py cfg2 gpurun_variants/lib_r6_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_cfg2.txt 2>&1; tail -4 $OUT/kbench_cfg2.txt
timeout 600 python tools/kbench.py cfg4 gpurun_variants/lib_r6_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_cfg4.txt 2>&1; tail -4 $OUT/kbench_cfg4.txt
cp gpurun_out/grad_margins.txt $OUT/ 2>/dev/null
