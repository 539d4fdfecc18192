#!/bin/bash
TAG=${1:-var2h}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 600 python tools/repeat_check.py cfg2 $V/lib_v2_d1.so,$V/lib_v2_d2.so,$V/lib_v2_d4.so,$V/lib_v2_d8.so --caps 0 --gemms bf16x3 --reps 8 > $OUT/repeat.txt 2>&1
grep distinct $OUT/repeat.txt
