#!/bin/bash
# soak run of the randomized parity tests (PINN_FUZZ_SCALE x the default number of cases) + the Solver.fit rates: bash tools/gpu_soak.sh <tag> [scale]
TAG=${1:-soak}; SCALE=${2:-6}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
PINN_FUZZ_SCALE=$SCALE timeout 1500 python -m pytest tests/test_fuzz_equations.py -m gpu -q --durations=6 > $OUT/fuzz_soak.txt 2>&1; echo "pytest exit $?" >> $OUT/fuzz_soak.txt
tail -n 12 $OUT/fuzz_soak.txt
cp gpurun_out/grad_margins.txt $OUT/grad_margins_soak.txt 2>/dev/null
timeout 500 python tools/fit_rate.py > $OUT/fit_rate.txt 2>&1; cat $OUT/fit_rate.txt
