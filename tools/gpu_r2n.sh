#!/bin/bash
# round 2, GPU session E: full parity suite, smoke, the four bench lines (with cpu_baseline) and their rocprof evidence
OUT=/root/repo/gpurun_out/r2n; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
for c in cfg2 cfg3 cfg4 cfg5; do timeout 600 python bench.py --workload $c > $OUT/bench_$c.txt 2> $OUT/bench_$c.err; tail -1 $OUT/bench_$c.err; done
timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --unfused > $OUT/bench_cfg4_unfused.txt 2> $OUT/bench_cfg4_unfused.err; grep "bench" $OUT/bench_cfg4_unfused.err | tail -1
for c in cfg2 cfg3 cfg4 cfg5; do timeout 900 bash tools/profile_bench.sh $c r2n > /dev/null 2>&1; tail -12 $OUT/prof_$c/summary.txt; done
timeout 400 python tools/fit_rate.py > $OUT/fit_rate.txt 2>&1; cat $OUT/fit_rate.txt
