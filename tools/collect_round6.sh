#!/bin/bash
# copy the judged evidence of tools/gpu_round6_final.sh from gpurun_out/<tag>/ into profiles/ (tracked): tools/collect_round6.sh <tag>
TAG=${1:-r6z}; SRC=gpurun_out/$TAG; cd /root/repo
for c in cfg2 cfg3 cfg4 cfg5; do
  for s in "" _split; do
    d=$SRC/prof_$c$s
    cp $d/summary.txt profiles/r06_${c}${s}_summary.txt
    cp $d/pmc.json profiles/r06_${c}${s}_pmc.json
    cp $d/kernel_stats.csv profiles/r06_${c}${s}_kernel_stats.csv
    tail -n 1 $SRC/bench_$c$s.txt > profiles/r06_${c}${s}_bench_line.txt
  done
done
for w in program sin64 sin128 generic skip128 skip256 gelu256 burgers64 heat64; do
  d=$SRC/prof_$w
  cp $d/summary.txt profiles/r06_breadth_${w}_summary.txt
  cp $d/pmc.json profiles/r06_breadth_${w}_pmc.json
  cp $d/kernel_stats.csv profiles/r06_breadth_${w}_kernel_stats.csv
  tail -n 1 $SRC/bench_$w.txt > profiles/r06_breadth_${w}_bench_line.txt
done
tail -n 1 $SRC/bench_default.txt > profiles/r06_default_bench_line.txt
cat $SRC/bench_default.time > profiles/r06_default_bench_wall_time.txt
tail -n 1 $SRC/bench_cfg2_driver_form.txt > profiles/r06_cfg2_driver_form_bench_line.txt
tail -n 1 $SRC/bench_cfg4_dp_path_n1.txt > profiles/r06_cfg4_dp_path_n1_bench_line.txt
tail -n 1 $SRC/bench_poisson512.txt > profiles/r06_breadth_poisson512_bench_line.txt
cp $SRC/wide512_rate.txt profiles/r06_wide512_rate_final.txt 2>/dev/null
cp $SRC/fit_rate.txt profiles/r06_fit_rate.txt
cp $SRC/small_fit_rate.txt profiles/r06_small_fit_rate.txt
cp $SRC/cfg4_bl_probe.txt profiles/r06_cfg4_bl_probe.txt
{ grep -E "passed|failed|exit|^[0-9.]+s " $SRC/pytest_gpu.log; } > profiles/r06_pytest_gpu.txt
cp $SRC/grad_margins.txt profiles/r06_grad_margins.txt 2>/dev/null
cp $SRC/smoke.txt profiles/r06_smoke.txt 2>/dev/null
