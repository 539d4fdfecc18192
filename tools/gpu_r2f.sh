#!/bin/bash
# round 2, GPU session F: full parity suite after third order / four directions; phases of the two-waves cfg4 kernel; fit rates
OUT=/root/repo/gpurun_out/r2f; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; grep -E "passed|failed|^FAILED" $OUT/pytest_gpu.log | tail -8
for c in cfg4 cfg2; do timeout 200 python tools/phases.py gpurun_variants/lib_phases.so $c > $OUT/phases_$c.txt 2>&1; cat $OUT/phases_$c.txt; done
timeout 400 python tools/fit_rate.py > $OUT/fit_rate.txt 2>&1; cat $OUT/fit_rate.txt
