#!/bin/bash
# the GPU parity suite + smoke at HEAD: tools/gpu_tests.sh [tag]  -> gpurun_out/<tag>/pytest_gpu.log
TAG=${1:-tests}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|Error|error|exit" $OUT/pytest_gpu.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
