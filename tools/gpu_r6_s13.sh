#!/bin/bash
# round 6 session 13: does the fp64 pre-pass decide between the one-CU fit chunk and the launch graphs on BASELINE config 1? + the poisson512 line with the per-launch roofline
TAG=${1:-r6s13}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 600 python tools/small_fit_rate.py cfg1,cfg1_nosrc > $OUT/small_fit_cfg1.txt 2>&1; cat $OUT/small_fit_cfg1.txt
timeout 400 python bench.py --workload poisson512 --no-cpu-baseline --no-strong > $OUT/bench_poisson512.txt 2> $OUT/bench_poisson512.err; grep 'bench\] gpu' $OUT/bench_poisson512.err
