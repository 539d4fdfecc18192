#!/bin/bash
# Round-6 evidence in ONE session (one box), on the FINAL sources: rocprofv3 kernel trace + PMC passes of every BASELINE workload in both
# GEMM modes AND of the breadth workloads (tools/profile_bench.sh: --no-parity --no-cold --no-configs, so that every launch in a kernel row
# has the workload's grid), THEN the bench lines (`roofline.traffic` = the PMC figure of the same session and kernel sources; the default
# line carries `baseline_configs` + `sustained`), the breadth lines, fit rates, the GPU suite last.
# usage: bash tools/gpu_round6_final.sh <tag>   -> gpurun_out/<tag>/ ; copy into profiles/ with tools/collect_round6.sh <tag>
TAG=${1:-r6z}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
BREADTH="program sin64 sin128 generic skip128 skip256 gelu256 burgers64 heat64"
for c in cfg2 cfg4 cfg3 cfg5; do
  timeout 420 bash tools/profile_bench.sh $c $TAG "" > /dev/null 2>&1
  timeout 420 bash tools/profile_bench.sh $c $TAG _split --gemm bf16x3 > /dev/null 2>&1
  cp $OUT/prof_$c/pmc.json profiles/r06_${c}_pmc.json 2>/dev/null
  cp $OUT/prof_${c}_split/pmc.json profiles/r06_${c}_split_pmc.json 2>/dev/null
done
for w in $BREADTH; do
  timeout 420 bash tools/profile_bench.sh $w $TAG "" > /dev/null 2>&1
  cp $OUT/prof_$w/pmc.json profiles/r06_breadth_${w}_pmc.json 2>/dev/null
done
( time timeout 600 python bench.py > $OUT/bench_default.txt 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "default bench rc=$? $(grep real $OUT/bench_default.time)"
for c in cfg2 cfg3 cfg4 cfg5; do
  timeout 400 python bench.py --workload $c --no-configs > $OUT/bench_$c.txt 2> $OUT/bench_$c.err; echo "$c fp32: $(grep 'bench\] gpu' $OUT/bench_$c.err)"
  timeout 300 python bench.py --workload $c --gemm bf16x3 --no-cpu-baseline --no-configs > $OUT/bench_${c}_split.txt 2> $OUT/bench_${c}_split.err; echo "$c bf16x3: $(grep 'bench\] gpu' $OUT/bench_${c}_split.err)"
done
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_cfg2_driver_form.txt 2> $OUT/bench_cfg2_driver_form.err
timeout 300 python bench.py --unfused --workload cfg4 --no-cpu-baseline --no-strong --no-side --no-configs > $OUT/bench_cfg4_dp_path_n1.txt 2> $OUT/bench_cfg4_dp_path_n1.err
echo "cfg4 DP path at one rank: $(grep 'bench\] gpu' $OUT/bench_cfg4_dp_path_n1.err)"
for w in $BREADTH; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-strong > $OUT/bench_$w.txt 2> $OUT/bench_$w.err
  echo "$w: $(grep 'bench\] gpu' $OUT/bench_$w.err) parity $(python -c "import json; print(json.loads(open('$OUT/bench_$w.txt').read().strip().splitlines()[-1]).get('parity_checked', {}).get('ok'))" 2>/dev/null)"
done
timeout 400 python bench.py --workload poisson512 --no-cpu-baseline --no-strong > $OUT/bench_poisson512.txt 2> $OUT/bench_poisson512.err; echo "poisson512: $(grep 'bench\] gpu' $OUT/bench_poisson512.err)"
timeout 300 python tools/wide512_rate.py > $OUT/wide512_rate.txt 2>&1; tail -n 3 $OUT/wide512_rate.txt
timeout 500 python tools/fit_rate.py > $OUT/fit_rate.txt 2>&1; tail -n 8 $OUT/fit_rate.txt
timeout 400 python tools/small_fit_rate.py > $OUT/small_fit_rate.txt 2>&1; cat $OUT/small_fit_rate.txt
timeout 600 python tools/cfg4_bl_probe.py 99 100 101 > $OUT/cfg4_bl_probe.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|exit" $OUT/pytest_gpu.log | tail -n 4
cp gpurun_out/grad_margins.txt $OUT/grad_margins.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
