#!/bin/bash
# rocprofv3 kernel stats + PMC passes for the breadth workloads, then their bench lines with `traffic` from this session
#   usage: bash tools/gpu_breadth_prof.sh <tag> <workloads...>  -> gpurun_out/<tag>/ and profiles/r04_breadth_<w>_{pmc.json,summary.txt,bench_line.txt}
TAG=$1; shift; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for w in "$@"; do
  timeout 300 bash tools/profile_bench.sh $w $TAG "" > /dev/null 2>&1
  cp $OUT/prof_$w/pmc.json profiles/r04_${w}_pmc.json 2>/dev/null
  cp $OUT/prof_$w/summary.txt $OUT/summary_$w.txt 2>/dev/null
  cp $OUT/prof_$w/kernel_stats.csv $OUT/kernel_stats_$w.csv 2>/dev/null
done
for w in "$@"; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-strong > $OUT/bench_$w.txt 2> $OUT/bench_$w.err
  echo "$w: $(grep 'bench\] gpu' $OUT/bench_$w.err) traffic $(python -c "import json; print(json.loads(open('$OUT/bench_$w.txt').read().strip().splitlines()[-1])['roofline'].get('traffic'))" 2>/dev/null)"
  cp profiles/r04_${w}_pmc.json $OUT/pmc_$w.json 2>/dev/null
done
