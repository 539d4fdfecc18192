#!/bin/bash
# Round-5 evidence in ONE session (one box), on the FINAL sources: rocprofv3 kernel trace + PMC passes of every BASELINE workload in both
# GEMM modes (tools/profile_bench.sh: --no-parity --no-cold, so that every launch in a kernel row has the workload's grid), THEN the bench
# lines (`roofline.traffic` = the PMC figure of the same session and kernel sources), the breadth lines, fit rates, the GPU suite last.
# usage: bash tools/gpu_round5_final.sh <tag>   -> gpurun_out/<tag>/ ; copy into profiles/ with tools/collect_round5.sh <tag>
TAG=${1:-r5z}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for c in cfg2 cfg4 cfg3 cfg5; do
  timeout 420 bash tools/profile_bench.sh $c $TAG "" > /dev/null 2>&1
  timeout 420 bash tools/profile_bench.sh $c $TAG _split --gemm bf16x3 > /dev/null 2>&1
  cp $OUT/prof_$c/pmc.json profiles/r05_${c}_pmc.json 2>/dev/null
  cp $OUT/prof_${c}_split/pmc.json profiles/r05_${c}_split_pmc.json 2>/dev/null
done
for c in cfg2 cfg3 cfg4 cfg5; do
  timeout 400 python bench.py --workload $c > $OUT/bench_$c.txt 2> $OUT/bench_$c.err; echo "$c fp32: $(grep 'bench\] gpu' $OUT/bench_$c.err)"
  timeout 300 python bench.py --workload $c --gemm bf16x3 --no-cpu-baseline > $OUT/bench_${c}_split.txt 2> $OUT/bench_${c}_split.err; echo "$c bf16x3: $(grep 'bench\] gpu' $OUT/bench_${c}_split.err)"
done
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_cfg2_driver_form.txt 2> $OUT/bench_cfg2_driver_form.err
timeout 300 python bench.py --unfused --workload cfg4 --no-cpu-baseline --no-strong --no-side > $OUT/bench_cfg4_dp_path_n1.txt 2> $OUT/bench_cfg4_dp_path_n1.err
echo "cfg4 DP path at one rank: $(grep 'bench\] gpu' $OUT/bench_cfg4_dp_path_n1.err)"
for w in skip128 skip256 sin64 sin128 gelu256 program generic; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-strong > $OUT/bench_$w.txt 2> $OUT/bench_$w.err
  echo "$w: $(grep 'bench\] gpu' $OUT/bench_$w.err) parity $(python -c "import json; print(json.loads(open('$OUT/bench_$w.txt').read().strip().splitlines()[-1]).get('parity_checked', {}).get('ok'))" 2>/dev/null)"
done
timeout 500 python tools/fit_rate.py > $OUT/fit_rate.txt 2>&1; tail -8 $OUT/fit_rate.txt
timeout 400 python tools/small_fit_rate.py > $OUT/small_fit_rate.txt 2>&1; cat $OUT/small_fit_rate.txt
if [ -n "$PYTEST_SUBSET" ]; then   # (a session short of GPU minutes: the suite without its slowest sweeps; the log says so)
  echo "# SUBSET: -k '$PYTEST_SUBSET' (the full suite ran on the previous commit: see the second block)" > $OUT/pytest_gpu.log
  timeout ${PYTEST_TIMEOUT:-150} python -m pytest tests -m gpu -q -k "$PYTEST_SUBSET" >> $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
else
  timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
fi
grep -E "passed|failed|exit" $OUT/pytest_gpu.log | tail -4
cp gpurun_out/grad_margins.txt $OUT/grad_margins.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
