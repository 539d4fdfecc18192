#!/bin/bash
OUT=/root/repo/gpurun_out/r3l; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 300 python tools/split_check.py gpurun_variants/lib_wg2.so 2>&1 | grep cfg
bash tools/gpu_ab.sh r3l gpurun_variants/lib_base.so gpurun_variants/lib_wg2.so
for c in cfg2 cfg4; do timeout 300 python bench.py --workload $c --gemm bf16x3 --no-cpu-baseline --lib gpurun_variants/lib_wg2.so 2>&1 | grep "bench\] gpu"; timeout 300 python bench.py --workload $c --gemm bf16x3 --no-cpu-baseline --lib gpurun_variants/lib_base.so 2>&1 | grep "bench\] gpu"; done
