#!/bin/bash
# round 5, GPU session 1: same-box A/B of the headline-kernel candidates (tools/variant.sh builds under gpurun_variants/), the new
# -m gpu tests of the round, the one-rank data-parallel step path.   usage: bash tools/gpu_r5_s1.sh <tag>
TAG=${1:-r5a}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
LIBS=$(ls gpurun_variants/lib_r5_*.so 2>/dev/null | tr '\n' ' ')
for c in cfg2 cfg4; do
  timeout 500 python tools/kbench.py $c $LIBS > $OUT/kbench_$c.txt 2>&1; cat $OUT/kbench_$c.txt
done
timeout 900 python -m pytest tests -m gpu -q -x --durations=8 \
  -k "alias or tutorial_script or layout_breadth or launch_graph or graph or occupancy or chunk or wide_residual" > $OUT/pytest_sel.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_sel.log; tail -15 $OUT/pytest_sel.log
timeout 300 python bench.py --unfused --workload cfg4 --no-cpu-baseline --no-strong --no-side > $OUT/dp_path_n1_cfg4.txt 2> $OUT/dp_path_n1_cfg4.err
grep 'bench\] gpu' $OUT/dp_path_n1_cfg4.err
timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --no-strong --no-side > $OUT/fused_cfg4.txt 2> $OUT/fused_cfg4.err
grep 'bench\] gpu' $OUT/fused_cfg4.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg2_driver_form.txt 2> $OUT/bench_cfg2_driver_form.err
grep 'bench\] gpu' $OUT/bench_cfg2_driver_form.err
