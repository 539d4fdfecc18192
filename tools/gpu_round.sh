#!/bin/bash
# One gpurun call that re-takes every round-level measurement at HEAD:
#   GPU parity tests, bench.py (plain and under rocprofv3 --kernel-trace --stats), PMC passes of the tile kernel,
#   Solver.fit rates of all BASELINE configs.  usage: tools/gpu_round.sh <tag>   (output: gpurun_out/<tag>/)
TAG=${1:-round}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 300 python bench.py > $OUT/bench_line.txt 2> $OUT/bench_err.txt; tail -2 $OUT/bench_err.txt; cut -c1-400 $OUT/bench_line.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python /root/repo/bench.py > $OUT/bench_under_rocprof.txt 2>&1)
find $OUT/bench_trace -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
head -4 $OUT/bench_kernel_stats.csv | cut -c1-200
timeout 600 bash tools/profile.sh cfg2 $TAG/prof_cfg2 > /dev/null 2>&1
tail -40 $OUT/prof_cfg2/summary.txt
timeout 300 python tools/fit_rate.py > $OUT/fit_rate.txt 2>&1; cat $OUT/fit_rate.txt
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
