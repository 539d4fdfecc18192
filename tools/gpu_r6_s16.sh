#!/bin/bash
# round 6 session 16: experiment -- the fence behind the in-kernel pre-pass dropped (the barrier behind the point staging orders the rows): fit rates of the small problems, BASELINE kernels
TAG=${1:-r6s16}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 600 python tools/small_fit_rate.py cfg1,cfg1_nosrc,poisson_10,heat_sigmoid > $OUT/small_fit_product.txt 2>&1; grep -v "^#" $OUT/small_fit_product.txt | head -20
SMALL_FIT_LIB=/root/repo/$V/lib_nofence.so timeout 600 python tools/small_fit_rate.py cfg1,cfg1_nosrc,poisson_10,heat_sigmoid > $OUT/small_fit_nofence.txt 2>&1; grep -v "^#" $OUT/small_fit_nofence.txt | head -20
timeout 600 python tools/kbench.py cfg2 pydens_amd/libpinn_hip.so $V/lib_nofence.so > $OUT/kbench_cfg2.txt 2>&1; tail -n 4 $OUT/kbench_cfg2.txt
