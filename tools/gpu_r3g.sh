#!/bin/bash
OUT=/root/repo/gpurun_out/r3g; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 600 python tools/grad_margins.py > $OUT/grad_margins.txt 2>&1; grep -v Warn $OUT/grad_margins.txt | cut -c1-200
for g in fp32 bf16x3; do for c in cfg2 cfg4; do
  timeout 300 python bench.py --workload $c --gemm $g --no-cpu-baseline > $OUT/bench_${c}_$g.txt 2> $OUT/bench_${c}_$g.err; grep "bench\] gpu" $OUT/bench_${c}_$g.err
done; done
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
