#!/bin/bash
# round 6 session 15: the fp64 pre-pass interpreter with its common operations in front and the last result forwarded in a register (lib_pp2.so) against the product
TAG=${1:-r6s15}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 600 python tools/small_fit_rate.py cfg1,cfg1_nosrc,poisson_10,ode_family > $OUT/small_fit_product.txt 2>&1; grep -v "^#\|PERSIST=1" $OUT/small_fit_product.txt | head -20
SMALL_FIT_LIB=/root/repo/$V/lib_pp2.so timeout 600 python tools/small_fit_rate.py cfg1,cfg1_nosrc,poisson_10,ode_family > $OUT/small_fit_pp2.txt 2>&1; grep -v "^#" $OUT/small_fit_pp2.txt | head -20
timeout 600 python tools/kbench.py cfg2 pydens_amd/libpinn_hip.so $V/lib_pp2.so > $OUT/kbench_cfg2.txt 2>&1; tail -n 4 $OUT/kbench_cfg2.txt
timeout 600 python tools/kbench.py cfg4 pydens_amd/libpinn_hip.so $V/lib_pp2.so > $OUT/kbench_cfg4.txt 2>&1; tail -n 4 $OUT/kbench_cfg4.txt
