#!/bin/bash
# round 6 session 25: the two-team kernels with the prologue's two chains on different teams (W^T staging | pre-pass) and the teams' sums merged in LDS, against the build before
TAG=${1:-r6s25}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 600 python tools/kbench.py cfg2 $V/lib_p_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_cfg2.txt 2>&1; tail -n 4 $OUT/kbench_cfg2.txt
timeout 600 python tools/kbench.py cfg4 $V/lib_p_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_cfg4.txt 2>&1; tail -n 4 $OUT/kbench_cfg4.txt
timeout 600 python tools/fixed_cost.py cfg2 > $OUT/fixed_cfg2.txt 2>&1; tail -n 1 $OUT/fixed_cfg2.txt
timeout 300 python bench.py --no-cpu-baseline --no-strong --no-configs > $OUT/bench_cfg2.txt 2> $OUT/bench_cfg2.err; grep 'bench\] gpu' $OUT/bench_cfg2.err
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_occupancy.py -m gpu -q -k "cfg1 or cfg2 or cfg4 or bitwise or repeat or bench_parity or program or sin or evolution or golden or tutorial or occupancy or large_batch" > $OUT/pytest_teams.txt 2>&1; tail -n 3 $OUT/pytest_teams.txt
