#!/bin/bash
# round 6 session 19: the static-activation Sin kernel at widths 128 and 256 against the oracle (new GPU tests), sin128 in the large-batch / repeat tests
TAG=${1:-r6s19}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_occupancy.py -m gpu -q -k "wide_sin or sin128" > $OUT/pytest_sin.txt 2>&1; tail -n 3 $OUT/pytest_sin.txt
