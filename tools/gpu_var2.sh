#!/bin/bash
# GPU session behind DESIGN.md's account of the two-workgroups-per-CU finding: hardware probes (private memory per wave, vmcnt
# ordering), the VAR 2 experiment builds of tools/var2.sh against the two-team product form, the scratch canary, then the large-batch
# tests of the product library. tools/gpu_var2.sh [tag] -> gpurun_out/<tag>/
TAG=${1:-var2}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 120 $V/ubench/scratch_alias > $OUT/scratch_alias.txt 2>&1; tail -3 $OUT/scratch_alias.txt
timeout 120 $V/ubench/vmcnt_order > $OUT/vmcnt_order.txt 2>&1; tail -3 $OUT/vmcnt_order.txt
timeout 600 python tools/repeat_check.py cfg2,cfg4 $V/lib_base.so,$V/lib_v2.so,$V/lib_v2_fz.so --caps 0,1 --gemms fp32,bf16x3 > $OUT/repeat.txt 2>&1
grep distinct $OUT/repeat.txt
PINN_CANARY=1 timeout 300 python tools/repeat_check.py cfg2,cfg4 $V/lib_v2_canary.so,$V/lib_base_canary.so --caps 0 --gemms fp32,bf16x3 > $OUT/canary.txt 2>&1
grep "distinct\|canary" $OUT/canary.txt
timeout 1500 python -m pytest tests/test_gpu_occupancy.py -m gpu -q -s --durations=10 > $OUT/pytest_occupancy.log 2>&1; tail -25 $OUT/pytest_occupancy.log
