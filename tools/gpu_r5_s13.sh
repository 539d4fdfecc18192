#!/bin/bash
# round 5, session 13: scheduler strategy of the width-128 / 256 translation units (BASELINE configs 3 / 5: tile + weight-gradient kernels), same-box A/B
TAG=${1:-r5t}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
L="gpurun_variants/lib_w_dflt.so gpurun_variants/lib_w_ilp.so gpurun_variants/lib_w_maxilp.so gpurun_variants/lib_w_maxocc.so"
timeout 500 python tools/kbench.py cfg3 $L > $OUT/kbench_cfg3.txt 2>&1; grep -v amdgpu $OUT/kbench_cfg3.txt | tail -8
timeout 500 python tools/kbench.py cfg5 $L > $OUT/kbench_cfg5.txt 2>&1; grep -v amdgpu $OUT/kbench_cfg5.txt | tail -8
