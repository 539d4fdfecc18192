#!/bin/bash
# copy the judged evidence of tools/gpu_round5_final.sh from gpurun_out/<tag>/ into profiles/ (tracked): tools/collect_round5.sh <tag>
TAG=${1:-r5z}; SRC=gpurun_out/$TAG; cd /root/repo
for c in cfg2 cfg3 cfg4 cfg5; do
  for s in "" _split; do
    d=$SRC/prof_$c$s
    cp $d/summary.txt profiles/r05_${c}${s}_summary.txt
    cp $d/pmc.json profiles/r05_${c}${s}_pmc.json
    cp $d/kernel_stats.csv profiles/r05_${c}${s}_kernel_stats.csv
    tail -1 $SRC/bench_$c$s.txt > profiles/r05_${c}${s}_bench_line.txt
  done
done
tail -1 $SRC/bench_cfg2_driver_form.txt > profiles/r05_cfg2_driver_form_bench_line.txt
tail -1 $SRC/bench_cfg4_dp_path_n1.txt > profiles/r05_cfg4_dp_path_n1_bench_line.txt
for w in skip128 skip256 sin64 sin128 gelu256 program generic; do tail -1 $SRC/bench_$w.txt > profiles/r05_breadth_${w}_bench_line.txt; done
cp $SRC/fit_rate.txt profiles/r05_fit_rate.txt
cp $SRC/small_fit_rate.txt profiles/r05_small_fit_rate.txt
{ grep -E "passed|failed|exit|^[0-9.]+s " $SRC/pytest_gpu.log; } > profiles/r05_pytest_gpu.txt
cp $SRC/grad_margins.txt profiles/r05_grad_margins.txt 2>/dev/null
cp $SRC/smoke.txt profiles/r05_smoke.txt 2>/dev/null
