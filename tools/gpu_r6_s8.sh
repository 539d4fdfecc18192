#!/bin/bash
# round 6 session 8: the two-team twins of the Poisson-box kernel for residual PROGRAMS (VAR 2048) and for the 'Sin' activation, against the kernels these workloads ran on
TAG=${1:-r6s8}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 600 python tools/kbench.py program $V/lib_p_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_program.txt 2>&1; tail -n 4 $OUT/kbench_program.txt
timeout 600 python tools/kbench.py sin64 $V/lib_p_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_sin64.txt 2>&1; tail -n 4 $OUT/kbench_sin64.txt
timeout 1500 python -m pytest tests/test_gpu_occupancy.py tests/test_gpu_parity.py -q -m gpu -k "occupancy or large_batch or sin_net or every_width or bitwise or launch_graph_policy or bench_parity or fuzz or tutorial or program or variable" > $OUT/pytest_sel.txt 2>&1; tail -n 6 $OUT/pytest_sel.txt
for w in program sin64; do timeout 300 python bench.py --workload $w --no-cpu-baseline --no-strong > $OUT/bench_$w.txt 2> $OUT/bench_$w.err; grep 'bench\] gpu' $OUT/bench_$w.err; done
