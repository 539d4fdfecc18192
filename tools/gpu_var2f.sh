#!/bin/bash
TAG=${1:-var2f}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 300 python tools/diff_runs.py cfg2 $V/lib_v2_dump.so --reps 4 --net --brief --save $OUT/lanes > $OUT/net_cfg2.txt 2>&1; cat $OUT/net_cfg2.txt
