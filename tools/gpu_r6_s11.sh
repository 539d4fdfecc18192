#!/bin/bash
# round 6 session 11: width 512 with the weight-gradient kernel's lambdas inlined; the wide (128 / 256) weight-gradient tests on the same sources
TAG=${1:-r6s11}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 600 python tools/wide512_rate.py > $OUT/wide512_rate.txt 2>&1; tail -n 4 $OUT/wide512_rate.txt
timeout 400 python bench.py --workload poisson512 --no-cpu-baseline --no-strong > $OUT/bench_poisson512.txt 2> $OUT/bench_poisson512.err; grep 'bench\] gpu' $OUT/bench_poisson512.err
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_occupancy.py -q -m gpu -k "hidden_width_512 or wgx or wide or skip or cfg3 or cfg5 or w100 or width_and_depth or breadth" > $OUT/pytest_wide.txt 2>&1; tail -n 5 $OUT/pytest_wide.txt
timeout 600 python tools/kbench.py cfg5 gpurun_variants/lib_p_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_cfg5.txt 2>&1; tail -n 4 $OUT/kbench_cfg5.txt
