#!/bin/bash
# fourth GPU session of the investigation: when is an MFMA result readable beside a busy partner wave (tools/ubench/simd_coexec.cpp);
# long repeat runs of the product kernels. tools/gpu_var2d.sh [tag]
TAG=${1:-var2d}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 600 $V/ubench/simd_coexec > $OUT/simd_coexec.txt 2>&1; cat $OUT/simd_coexec.txt
timeout 600 python tools/repeat_check.py cfg2,cfg4 pydens_amd/libpinn_hip.so --caps 0 --gemms fp32,bf16x3 --reps 40 > $OUT/repeat_product.txt 2>&1
grep distinct $OUT/repeat_product.txt
