#!/bin/bash
# round 5, session 8: parity tests and rates of the one-launch fit chunk forms (product build)
TAG=${1:-r5k}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "fit_chunk or launch_graphs" > $OUT/pytest_fit.log 2>&1; tail -3 $OUT/pytest_fit.log
timeout 500 python tools/small_fit_rate.py > $OUT/small_fit_rate.txt 2>&1; cat $OUT/small_fit_rate.txt
