#!/bin/bash
# (a) where the skip128 tile kernel spends its cycles; (b) do the wide nets run faster when a chunk's slabs fit the 256 MB Infinity Cache?
TAG=${1:-chunks}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 300 python tools/phases.py gpurun_variants/lib_phases128.so skip128 > $OUT/phases_skip128.txt 2>&1; cat $OUT/phases_skip128.txt | tail -20
timeout 300 python tools/phases.py gpurun_variants/lib_phases128.so cfg3 > $OUT/phases_cfg3.txt 2>&1; cat $OUT/phases_cfg3.txt | tail -20
for w in cfg3 cfg5; do
  for mb in 0 2048 512 192 96; do
    export PYDENS_AMD_WGX_CHUNK_MB=$mb; [ $mb = 0 ] && unset PYDENS_AMD_WGX_CHUNK_MB
    timeout 300 python bench.py --workload $w --no-cpu-baseline --no-strong --no-side --steps 8 --warmup 2 > $OUT/bench_${w}_$mb.txt 2> $OUT/bench_${w}_$mb.err
    echo "$w chunk $mb MB: $(grep 'bench\] gpu' $OUT/bench_${w}_$mb.err)"
  done
done
unset PYDENS_AMD_WGX_CHUNK_MB
cd /tmp
for mb in 192; do
  PYDENS_AMD_WGX_CHUNK_MB=$mb rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg3_$mb -- python /root/repo/bench.py --workload cfg3 --no-cpu-baseline --no-strong --no-side --steps 8 --warmup 2 > $OUT/trace_cfg3_$mb.log 2>&1
  find $OUT/trace_cfg3_$mb -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_cfg3_$mb.csv \;
  rm -rf $OUT/trace_cfg3_$mb
  head -8 $OUT/kernel_stats_cfg3_$mb.csv
done
