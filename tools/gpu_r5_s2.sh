#!/bin/bash
# round 5, GPU session 2: the one-launch fit chunk (bit-identity tests first: a wrong device-scope wait shows there, bounded), mixed
# third-order partials, the soak of the split kernels behind the asm guard, small-batch fit rates, the headline kernel after PTALL + polynomial tanh
TAG=${1:-r5b}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x --durations=8 -k "one_launch or fit_chunks_as_launch_graphs or tiny_and_boundary" > $OUT/pytest_persist.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_persist.log; tail -6 $OUT/pytest_persist.log
timeout 900 python -m pytest tests -m gpu -q --durations=8 -k "direction_groups or golden or tutorial" > $OUT/pytest_sel.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_sel.log; tail -8 $OUT/pytest_sel.log
timeout 600 python tools/small_fit_rate.py > $OUT/small_fit_rate.txt 2>&1; cat $OUT/small_fit_rate.txt
timeout 300 python tools/kbench.py cfg2 gpurun_variants/lib_r5c_base.so gpurun_variants/lib_r5c_ptall.so gpurun_variants/lib_r5c_product.so > $OUT/kbench_cfg2.txt 2>&1; grep -v amdgpu $OUT/kbench_cfg2.txt
for g in fp32 bf16x3; do
  timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-strong --no-side --gemm $g > $OUT/bench_cfg2_$g.txt 2> $OUT/bench_cfg2_$g.err; echo "cfg2 $g: $(grep 'bench\] gpu' $OUT/bench_cfg2_$g.err)"
done
timeout 300 python bench.py --workload gelu256 --no-cpu-baseline --no-strong > $OUT/bench_gelu256.txt 2> $OUT/bench_gelu256.err; echo "gelu256: $(grep 'bench\] gpu' $OUT/bench_gelu256.err)"
