#!/bin/bash
# breadth workloads: bench lines + rocprofv3 kernel stats.   usage: tools/gpu_breadth.sh <tag>
TAG=${1:-breadth}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for w in skip128 sin64 program generic; do
  timeout 240 python bench.py --workload $w --no-strong > $OUT/bench_$w.txt 2> $OUT/bench_$w.err; grep "bench\] gpu" $OUT/bench_$w.err
  python - $OUT/bench_$w.txt <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d['roofline']
    print('   ', d['config']['workload'][:40], '| %.4g points/s | %s | alg frac %.3f executed frac %.3f' % (d['value'], r['kernel'], r['frac'], r['executed']['frac']))
except Exception as e:
    print('    no line:', e)
PY
done
