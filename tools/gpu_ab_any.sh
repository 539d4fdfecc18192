#!/bin/bash
# same-box A/B of builds on any workloads: tools/gpu_ab_any.sh <tag> "<workloads>" <libA.so> [...]   (the product library runs last in each round)
TAG=$1; WL=$2; shift 2; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for w in $WL; do
  for rep in 1 2; do
    for lib in "$@" product; do
      L=""; [ $lib != product ] && L="--lib $lib"
      timeout 300 python bench.py --workload $w --no-cpu-baseline --no-strong --no-side $L > $OUT/bench_${w}_$(basename $lib .so)_$rep.txt 2> $OUT/bench_${w}_$(basename $lib .so)_$rep.err
      echo "$w $(basename $lib .so) #$rep: $(grep 'bench\] gpu' $OUT/bench_${w}_$(basename $lib .so)_$rep.err)"
    done
  done
done
