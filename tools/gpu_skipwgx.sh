#!/bin/bash
# streamed weight gradients of the skip kernels (round 4): parity at widths 128 / 256, then the skip128 / skip256 bench lines
TAG=${1:-skipwgx}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_occupancy.py tests/test_fuzz_equations.py -m gpu -q -x --durations=8 \
  -k "wide_residual or skip128 or skip256 or layout or wide or w100" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
for w in skip128 skip256; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --no-strong > $OUT/bench_$w.txt 2> $OUT/bench_$w.err; echo "$w: $(grep 'bench\] gpu' $OUT/bench_$w.err)"; tail -c 1500 $OUT/bench_$w.txt
done
