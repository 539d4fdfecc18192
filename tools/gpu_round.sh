#!/bin/bash
# One gpurun call that re-takes every round-level measurement at HEAD: full GPU parity suite, smoke, the four bench lines
# (with cpu_baseline) and their rocprofv3 evidence (kernel stats + PMC passes), the data-parallel step path on a one-rank
# RCCL group, Solver.fit rates.   usage: tools/gpu_round.sh [tag]   (output: gpurun_out/<tag>/; copy what is to be judged
# into profiles/ as rNN_*)
TAG=${1:-round}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
for c in cfg2 cfg3 cfg4 cfg5; do timeout 600 python bench.py --workload $c > $OUT/bench_$c.txt 2> $OUT/bench_$c.err; tail -1 $OUT/bench_$c.err; done
timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --unfused > $OUT/bench_cfg4_unfused.txt 2> $OUT/bench_cfg4_unfused.err; grep "bench" $OUT/bench_cfg4_unfused.err | tail -1
for c in cfg2 cfg3 cfg4 cfg5; do timeout 900 bash tools/profile_bench.sh $c $TAG > /dev/null 2>&1; tail -12 $OUT/prof_$c/summary.txt; done
timeout 400 python tools/fit_rate.py > $OUT/fit_rate.txt 2>&1; cat $OUT/fit_rate.txt
