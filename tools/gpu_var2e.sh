#!/bin/bash
# where do two runs differ? tools/gpu_var2e.sh [tag]
TAG=${1:-var2e}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 300 python tools/diff_runs.py cfg2 $V/lib_v2.so --reps 5 > $OUT/diff_cfg2.txt 2>&1; cat $OUT/diff_cfg2.txt
timeout 300 python tools/diff_runs.py cfg4 $V/lib_v2_slpswap.so --reps 5 > $OUT/diff_cfg4.txt 2>&1; cat $OUT/diff_cfg4.txt
