#!/bin/bash
# Round-4 evidence in ONE session (one box): rocprofv3 kernel trace + PMC passes of every BASELINE workload in both GEMM modes, THEN the
# bench lines (so that `roofline.traffic` of a line is the PMC figure of the same session and the same kernel sources), the breadth lines,
# the fit rates.   usage: bash tools/gpu_round4_final.sh <tag>   -> gpurun_out/<tag>/ ; copy into profiles/ with tools/collect_round4.sh <tag>
TAG=${1:-r4z}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for c in cfg2 cfg4 cfg3 cfg5; do
  timeout 420 bash tools/profile_bench.sh $c $TAG "" > /dev/null 2>&1
  timeout 420 bash tools/profile_bench.sh $c $TAG _split --gemm bf16x3 > /dev/null 2>&1
  cp $OUT/prof_$c/pmc.json profiles/r04_${c}_pmc.json 2>/dev/null
  cp $OUT/prof_${c}_split/pmc.json profiles/r04_${c}_split_pmc.json 2>/dev/null
done
for c in cfg2 cfg3 cfg4 cfg5; do
  timeout 400 python bench.py --workload $c > $OUT/bench_$c.txt 2> $OUT/bench_$c.err; echo "$c fp32: $(grep 'bench\] gpu' $OUT/bench_$c.err)"
  timeout 300 python bench.py --workload $c --gemm bf16x3 --no-cpu-baseline > $OUT/bench_${c}_split.txt 2> $OUT/bench_${c}_split.err; echo "$c bf16x3: $(grep 'bench\] gpu' $OUT/bench_${c}_split.err)"
done
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_cfg2_driver_form.txt 2> $OUT/bench_cfg2_driver_form.err
for w in skip128 skip256 sin64 sin128 gelu256 program generic; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-strong > $OUT/bench_$w.txt 2> $OUT/bench_$w.err
  echo "$w: $(grep 'bench\] gpu' $OUT/bench_$w.err) parity $(python -c "import json; print(json.loads(open('$OUT/bench_$w.txt').read().strip().splitlines()[-1]).get('parity_checked', {}).get('ok'))" 2>/dev/null)"
done
for w in skip128 skip256 sin128 gelu256; do
  (cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/breadth_$w -- python /root/repo/bench.py --workload $w --no-cpu-baseline --no-strong --steps 30 --warmup 5 > $OUT/breadth_$w.log 2>&1)
  find $OUT/breadth_$w -name "*kernel_stats.csv" -exec cp {} $OUT/breadth_${w}_kernel_stats.csv \;
  rm -rf $OUT/breadth_$w
done
timeout 500 python tools/fit_rate.py > $OUT/fit_rate.txt 2>&1; tail -8 $OUT/fit_rate.txt
