#!/bin/bash
# round 6 session 18: the static-activation Sin kernel of width 128 against the full breadth kernel it replaces on that shape
TAG=${1:-r6s18}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 600 python tools/kbench.py sin128 gpurun_variants/lib_p_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_sin128.txt 2>&1; tail -n 4 $OUT/kbench_sin128.txt
timeout 300 python bench.py --workload sin128 --no-cpu-baseline --no-strong > $OUT/bench_sin128.txt 2> $OUT/bench_sin128.err; grep 'bench\] gpu' $OUT/bench_sin128.err
timeout 900 python -m pytest tests/test_gpu_occupancy.py tests/test_gpu_parity.py -m gpu -q -k "sin128 or sin or breadth or every_width" > $OUT/pytest_sin.txt 2>&1; tail -n 3 $OUT/pytest_sin.txt
