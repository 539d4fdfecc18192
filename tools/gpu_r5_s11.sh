#!/bin/bash
# round 5, session 11: scheduler strategy of the width-64 translation unit after the round's kernel changes (same-box A/B, baseline-only builds)
TAG=${1:-r5r}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
L="gpurun_variants/lib_r5x_ilp.so gpurun_variants/lib_r5x_dflt.so gpurun_variants/lib_r5x_maxocc.so gpurun_variants/lib_r5x_maxilp.so"
timeout 400 python tools/kbench.py cfg2 $L > $OUT/kbench_cfg2.txt 2>&1; grep -v amdgpu $OUT/kbench_cfg2.txt | tail -8
timeout 400 python tools/kbench.py cfg4 $L > $OUT/kbench_cfg4.txt 2>&1; grep -v amdgpu $OUT/kbench_cfg4.txt | tail -8
