#!/bin/bash
# round 5, session 4: the lane-split point stage of the 32-point-tile kernels (config 4 and its split twin), same-box A/B
TAG=${1:-r5d}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 400 python tools/kbench.py cfg4 gpurun_variants/lib_r5d_16only.so gpurun_variants/lib_r5d_split.so > $OUT/kbench_cfg4.txt 2>&1; grep -v amdgpu $OUT/kbench_cfg4.txt
timeout 300 python tools/kbench.py cfg2 gpurun_variants/lib_r5d_16only.so gpurun_variants/lib_r5d_split.so > $OUT/kbench_cfg2.txt 2>&1; grep -v amdgpu $OUT/kbench_cfg2.txt
for l in 16only split; do
  timeout 200 python bench.py --workload cfg4 --gemm bf16x3 --no-cpu-baseline --no-strong --no-parity --lib gpurun_variants/lib_r5d_$l.so > $OUT/bench_cfg4_split_$l.txt 2> $OUT/bench_cfg4_split_$l.err; echo "cfg4 bf16x3 $l: $(grep 'bench\] gpu' $OUT/bench_cfg4_split_$l.err)"
done
