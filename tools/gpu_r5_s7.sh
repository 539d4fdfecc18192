#!/bin/bash
# round 5, session 7: phase clocks of the one-CU fit chunk (-DPINN_FIT_PROF build of the width-16 kernels)
TAG=${1:-r5h}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for c in cfg1 cfg1_256 ode_tanh; do
  SMALL_FIT_LIB=/root/repo/gpurun_variants/lib_fitprof.so PYDENS_AMD_FIT_PERSIST=2 PYDENS_AMD_FIT_ROUNDS=1000 timeout 200 python tools/small_fit_rate.py $c 2560 > $OUT/fitprof_$c.txt 2>&1
  grep -v amdgpu $OUT/fitprof_$c.txt | tail -5
done
