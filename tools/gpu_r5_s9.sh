#!/bin/bash
# round 5, session 9: same-box A/B of the headline kernels before / after the epilogue + pre-pass changes (baseline-only builds)
TAG=${1:-r5l}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 300 python tools/kbench.py cfg2 gpurun_variants/lib_r5x_old.so gpurun_variants/lib_r5x_new.so gpurun_variants/lib_r5x_u1.so > $OUT/kbench_cfg2.txt 2>&1; grep -v amdgpu $OUT/kbench_cfg2.txt | tail -8
timeout 300 python tools/kbench.py cfg4 gpurun_variants/lib_r5x_old.so gpurun_variants/lib_r5x_new.so gpurun_variants/lib_r5x_u1.so > $OUT/kbench_cfg4.txt 2>&1; grep -v amdgpu $OUT/kbench_cfg4.txt | tail -8
