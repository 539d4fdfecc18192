#!/bin/bash
# round 5, session 16: the reduction + Adam launch with its loads batched (step time beside the tile kernel's) + the bitwise tests that touch it
TAG=${1:-r5y2}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for c in cfg2 cfg4; do timeout 200 python bench.py --workload $c --no-cpu-baseline --no-strong --no-parity --no-cold > $OUT/b_$c.txt 2> $OUT/b_$c.err; echo "$c: $(grep 'bench\] gpu' $OUT/b_$c.err)"; done
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "launch_graphs or golden" > $OUT/pytest_sub.log 2>&1; tail -2 $OUT/pytest_sub.log
