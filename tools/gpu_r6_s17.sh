#!/bin/bash
# round 6 session 17: fuzz soak (6 x the default number of random equations / layouts / shapes) and 1 000-launch bitwise soaks on the final sources
TAG=${1:-r6s17}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
PINN_FUZZ_SCALE=6 timeout 1800 python -m pytest tests/test_fuzz_equations.py -m gpu -q --durations=6 > $OUT/fuzz_soak.txt 2>&1; echo "pytest exit $?" >> $OUT/fuzz_soak.txt; tail -n 12 $OUT/fuzz_soak.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_occupancy.py -m gpu -q -k "soak or bitwise or repeat or large_batch" > $OUT/repeat.txt 2>&1; tail -n 3 $OUT/repeat.txt
