#!/bin/bash
OUT=/root/repo/gpurun_out/r2m; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; grep -E "passed|failed|^FAILED" $OUT/pytest_gpu.log | tail -5
for c in cfg2 cfg4; do timeout 300 python bench.py --workload $c --no-cpu-baseline > $OUT/bench_$c.txt 2> $OUT/bench_$c.err; tail -1 $OUT/bench_$c.err; done
