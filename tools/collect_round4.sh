#!/bin/bash
# copy the judged evidence of tools/gpu_round4_final.sh from gpurun_out/<tag>/ into profiles/ (tracked): tools/collect_round4.sh <tag>
TAG=${1:-r4z}; SRC=gpurun_out/$TAG; cd /root/repo
for c in cfg2 cfg3 cfg4 cfg5; do
  for s in "" _split; do
    d=$SRC/prof_$c$s
    cp $d/summary.txt profiles/r04_${c}${s}_summary.txt
    cp $d/pmc.json profiles/r04_${c}${s}_pmc.json
    cp $d/kernel_stats.csv profiles/r04_${c}${s}_kernel_stats.csv
    tail -1 $SRC/bench_$c$s.txt > profiles/r04_${c}${s}_bench_line.txt
  done
done
tail -1 $SRC/bench_cfg2_driver_form.txt > profiles/r04_cfg2_driver_form_bench_line.txt
for w in skip128 skip256 sin64 sin128 gelu256 program generic; do tail -1 $SRC/bench_$w.txt > profiles/r04_breadth_${w}_bench_line.txt; done
for w in skip128 skip256 sin128 gelu256; do cp $SRC/breadth_${w}_kernel_stats.csv profiles/r04_breadth_${w}_kernel_stats.csv; done
cp $SRC/fit_rate.txt profiles/r04_fit_rate.txt
