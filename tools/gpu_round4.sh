#!/bin/bash
# Round-4 evidence in ONE session: the GPU suite + smoke, the default bench line (driver form and long form), the breadth lines.
#   usage: bash tools/gpu_round4.sh <tag>  -> gpurun_out/<tag>/
TAG=${1:-r4a}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|Error|error|exit" $OUT/pytest_gpu.log | tail -8
cp gpurun_out/grad_margins.txt $OUT/grad_margins.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_cfg2_driver_form.txt 2> $OUT/bench_cfg2_driver_form.err; tail -c 600 $OUT/bench_cfg2_driver_form.txt
timeout 400 python bench.py > $OUT/bench_cfg2.txt 2> $OUT/bench_cfg2.err; grep 'bench\] gpu' $OUT/bench_cfg2.err
for w in skip128 sin64 program generic; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-strong > $OUT/bench_$w.txt 2> $OUT/bench_$w.err; echo "$w: $(grep 'bench\] gpu' $OUT/bench_$w.err) parity $(python -c "import json,sys; print(json.loads(open('$OUT/bench_$w.txt').read().strip().splitlines()[-1]).get('parity_checked'))" 2>/dev/null)"
done
