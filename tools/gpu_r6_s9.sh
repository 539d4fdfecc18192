#!/bin/bash
# round 6 session 9: hidden width 512 and the (x, t) evolution shape on two-team kernels, on the device
TAG=${1:-r6s9}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "hidden_width_512 or evolution_shape" > $OUT/pytest_new.txt 2>&1; tail -n 6 $OUT/pytest_new.txt
timeout 600 python tools/wide512_rate.py > $OUT/wide512_rate.txt 2>&1; tail -n 4 $OUT/wide512_rate.txt
