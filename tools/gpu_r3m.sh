#!/bin/bash
OUT=/root/repo/gpurun_out/r3m; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
bash tools/gpu_tests.sh r3m
for c in cfg2 cfg4; do timeout 900 bash tools/profile_bench.sh $c r3m _split --gemm bf16x3 > /dev/null 2>&1; tail -42 $OUT/prof_${c}_split/summary.txt; done
