#!/bin/bash
# launch-graph replay of fit chunks: the bit-for-bit test, then the fit rates with and without it. tools/gpu_graph.sh [tag]
TAG=${1:-graph}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "launch_graphs" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for g in 0 1; do for c in cfg1 cfg1; do PYDENS_AMD_FIT_GRAPH=$g timeout 200 python tools/fit_one.py $c 6000 2>&1 | grep "^cfg" | sed "s/^/graph=$g /" | tee -a $OUT/fit_rate.txt; done; done
for g in 0 1; do PYDENS_AMD_FIT_GRAPH=$g timeout 200 python tools/fit_one.py cfg2 300 2>&1 | grep "^cfg" | sed "s/^/graph=$g /" | tee -a $OUT/fit_rate.txt; done
