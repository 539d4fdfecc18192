#!/bin/bash
# all-or-nothing adoption of the weight-gradient kernel's new epilogue addressing: the tests that run pinn_wgrad_kernel, then the PMC
# passes + kernel stats of the default workload on the new sources, then the driver-form bench line
OUT=/root/repo/gpurun_out/wge; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide_nets_at_their_full_batch or bitwise_repeatable or wide_residual" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
grep -E "passed|failed|exit" $OUT/pytest.log | tail -2
timeout 400 bash tools/profile_bench.sh cfg2 wge "" > /dev/null 2>&1
cp $OUT/prof_cfg2/pmc.json profiles/r04_cfg2_pmc.json; cp $OUT/prof_cfg2/pmc.json $OUT/cfg2_pmc.json; cp $OUT/prof_cfg2/summary.txt $OUT/cfg2_summary.txt; cp $OUT/prof_cfg2/kernel_stats.csv $OUT/cfg2_kernel_stats.csv
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_cfg2_driver_form.txt 2> $OUT/bench_cfg2_driver_form.err
grep 'bench\] gpu' $OUT/bench_cfg2_driver_form.err; python -c "import json; d=json.loads(open('$OUT/bench_cfg2_driver_form.txt').read().strip().splitlines()[-1]); print('traffic', d['roofline']['traffic'], d['roofline']['traffic_source'])"
