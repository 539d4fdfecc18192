#!/bin/bash
# the generic step as a launch graph: parity tests, then the bench line eager vs replayed (same box)
TAG=${1:-gg}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "generic" --durations=5 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -12 $OUT/pytest.log
for g in 0 1 0 1; do
  PYDENS_AMD_STEP_GRAPH=$g timeout 300 python bench.py --workload generic --no-cpu-baseline --no-strong > $OUT/bench_generic_g$g.txt 2> $OUT/bench_generic_g$g.err
  echo "graph=$g: $(grep 'bench\] gpu' $OUT/bench_generic_g$g.err) $(python -c "import json; d=json.loads(open('$OUT/bench_generic_g$g.txt').read().strip().splitlines()[-1]); print(d['config']['step_path'][-60:], d['parity_checked']['ok'])")"
done
