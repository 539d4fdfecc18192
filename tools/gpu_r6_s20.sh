#!/bin/bash
# round 6 session 20: the full breadth kernel with GELU fixed at compile time (gelu256 / a 128-wide relative) against the run-time activation switch
TAG=${1:-r6s20}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 900 python tools/kbench.py gelu256 gpurun_variants/lib_p_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_gelu256.txt 2>&1; tail -n 4 $OUT/kbench_gelu256.txt
timeout 300 python bench.py --workload gelu256 --no-cpu-baseline --no-strong > $OUT/bench_gelu256.txt 2> $OUT/bench_gelu256.err; grep 'bench\] gpu' $OUT/bench_gelu256.err
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_occupancy.py -m gpu -q -k "breadth or gelu or layout or activation or skip" > $OUT/pytest_gelu.txt 2>&1; tail -n 3 $OUT/pytest_gelu.txt
