#!/bin/bash
# round 6 session 23: what a launch of the BASELINE width-64 kernels costs before and after its tile loop (kernel time against tile rounds per team)
TAG=${1:-r6s23}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 600 python tools/fixed_cost.py cfg2 > $OUT/fixed_cfg2.txt 2>&1; grep -v amdgpu $OUT/fixed_cfg2.txt
timeout 600 python tools/fixed_cost.py cfg4 > $OUT/fixed_cfg4.txt 2>&1; grep -v amdgpu $OUT/fixed_cfg4.txt
