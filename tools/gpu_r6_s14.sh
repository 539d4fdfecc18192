#!/bin/bash
# round 6 session 14: phase clocks of the one-CU fit chunk with the fp64 pre-pass (-DPINN_FIT_PROF build), BASELINE config 1 with and without its source term
TAG=${1:-r6s14}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for c in cfg1 cfg1_nosrc; do
  SMALL_FIT_LIB=/root/repo/gpurun_variants/lib_fitprof.so PYDENS_AMD_FIT_PERSIST=2 timeout 200 python tools/small_fit_rate.py $c 2560 > $OUT/fitprof_$c.txt 2>&1
  grep -v amdgpu $OUT/fitprof_$c.txt | tail -n 5
done
