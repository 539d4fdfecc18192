#!/bin/bash
# round 5, session 14: scheduler strategy of the width-64 split-bf16 units (BASELINE configs 2 / 4 under --gemm bf16x3), same box, two passes
TAG=${1:-r5u}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for pass in 1 2; do for c in cfg2 cfg4; do for l in dflt ilp maxilp maxocc; do
  timeout 200 python bench.py --workload $c --gemm bf16x3 --no-cpu-baseline --no-strong --no-parity --no-cold --lib gpurun_variants/lib_spl_$l.so > $OUT/b_${c}_${l}_$pass.txt 2> $OUT/b_${c}_${l}_$pass.err
  echo "$c bf16x3 $l pass $pass: $(grep 'bench\] gpu' $OUT/b_${c}_${l}_$pass.err)"
done; done; done
