#!/bin/bash
# round 6 session 24: clock ticks of workgroup 0 / thread 0 before, inside and behind the tile loop of BASELINE config 2's kernel (-DPINN_FIT_PROF marks of pinn_tile_body)
TAG=${1:-r6s24}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for c in cfg2 cfg4; do timeout 300 python tools/kbench.py --flags=77 $c gpurun_variants/lib_tprof.so > $OUT/tileprof_$c.txt 2>&1; grep tileprof $OUT/tileprof_$c.txt | tail -n 4; tail -n 1 $OUT/tileprof_$c.txt; done
