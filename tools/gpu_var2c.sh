#!/bin/bash
# third GPU session of the two-workgroups-per-CU investigation: packed fp32 VALU beside another wave's MFMAs (probe), the split kernels
# with the SLP vectoriser swapped (config 2 without packed ops, config 4 with them). tools/gpu_var2c.sh [tag]
TAG=${1:-var2c}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 300 $V/ubench/pk_mfma_coexec > $OUT/pk_mfma_coexec.txt 2>&1; cat $OUT/pk_mfma_coexec.txt
timeout 600 python tools/repeat_check.py cfg2,cfg4 $V/lib_v2_slpswap.so,$V/lib_v2.so --caps 0 --gemms bf16x3 --reps 8 > $OUT/repeat.txt 2>&1
grep -A8 distinct $OUT/repeat.txt
