#!/bin/bash
# round 5, session 10: after the tracer fix (callable IC beside a parameter column lowered again): tutorial problems, alias scripts, small-batch rates
TAG=${1:-r5q}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pydens_alias.py tests/test_tutorial_problems.py tests/test_gpu_parity.py -m gpu -q -x -k "tutorial or alias or script or callable or heat or notebook or fit_chunk" > $OUT/pytest_tut.log 2>&1; tail -4 $OUT/pytest_tut.log
timeout 500 python tools/small_fit_rate.py > $OUT/small_fit_rate.txt 2>&1; cat $OUT/small_fit_rate.txt
