#!/bin/bash
# round 3, first GPU session: split-bf16 feasibility (tools/ubench/split_bf16.cpp), thread-trace attempt on the cfg4 kernel,
# this box's baseline kernel times.   usage: gpurun -- bash tools/gpu_r3a.sh   -> gpurun_out/r3a/
OUT=/root/repo/gpurun_out/r3a; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 120 gpurun_variants/split_bf16 > $OUT/split_bf16.txt 2>&1; echo "ubench exit $?" >> $OUT/split_bf16.txt
timeout 200 python tools/kbench.py cfg2 > $OUT/kbench_cfg2.txt 2>&1
timeout 200 python tools/kbench.py cfg4 > $OUT/kbench_cfg4.txt 2>&1
cd /tmp
timeout 240 rocprofv3 --att --kernel-include-regex "pinn_tile_kernel" --att-target-cu 1 --att-consecutive-kernels 2 -d $OUT/att -- \
    python /root/repo/bench.py --workload cfg4 --no-cpu-baseline --steps 3 --warmup 1 --settle 0 > $OUT/att.log 2>&1
echo "att exit $?" >> $OUT/att.log
cd /root/repo
find $OUT/att -type f | head -50 > $OUT/att_files.txt; du -sh $OUT/att >> $OUT/att_files.txt 2>&1
find $OUT/att -type f -size +4M -delete
tail -5 $OUT/split_bf16.txt; tail -3 $OUT/kbench_cfg2.txt $OUT/kbench_cfg4.txt; tail -15 $OUT/att.log
