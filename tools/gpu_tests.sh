#!/bin/bash
# the GPU suite + smoke (+ what the evidence session left to redo): bash tools/gpu_tests.sh <tag> [workloads whose bench line to take again]
TAG=${1:-tests}; shift; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|Error|error|exit" $OUT/pytest_gpu.log | tail -8
cp gpurun_out/grad_margins.txt $OUT/grad_margins.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
for w in "$@"; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-strong > $OUT/bench_$w.txt 2> $OUT/bench_$w.err; echo "$w: $(grep 'bench\] gpu' $OUT/bench_$w.err)"
done
timeout 500 python tools/fit_rate.py > $OUT/fit_rate.txt 2>&1; cat $OUT/fit_rate.txt
