#!/bin/bash
# second GPU session of the two-workgroups-per-CU investigation: MFMA / LDS hazard probe, the operand-prefetch forms and the
# "every fragment load behind the completion of the MFMAs before it" build of the config 2 split kernel. tools/gpu_var2b.sh [tag]
TAG=${1:-var2b}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 300 $V/ubench/mfma_lds_hazard > $OUT/mfma_lds_hazard.txt 2>&1; cat $OUT/mfma_lds_hazard.txt
timeout 600 python tools/repeat_check.py cfg2 $V/lib_v2.so,$V/lib_v2_pipe11.so,$V/lib_v2_pipe00.so,$V/lib_v2_il0.so,$V/lib_v2_drain.so --caps 0,1 --gemms bf16x3 --reps 8 > $OUT/repeat.txt 2>&1
grep -A8 distinct $OUT/repeat.txt
timeout 1500 python -m pytest tests/test_gpu_occupancy.py -m gpu -q -s --durations=10 > $OUT/pytest_occupancy.log 2>&1; grep -v "^$" $OUT/pytest_occupancy.log | tail -30
