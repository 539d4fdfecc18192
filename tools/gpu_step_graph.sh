#!/bin/bash
# the gradient part of an eager iteration as a launch graph (generic step; fused equation + constraint terms): parity tests, then the rates
TAG=${1:-sg}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "launch_graph or constraint or variable or tutorial or notebook" --durations=5 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -12 $OUT/pytest.log
for g in 0 1; do echo "PYDENS_AMD_STEP_GRAPH=$g"; PYDENS_AMD_STEP_GRAPH=$g timeout 300 python tools/generic_rate.py 2>&1 | grep batch; done | tee $OUT/generic_rate.txt
