#!/bin/bash
# round 2, GPU session C: parity tests, k2 prefetch A/B + memory-latency experiments, cfg4 two-waves A/B, fit rates
OUT=/root/repo/gpurun_out/r2c; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 300 python tools/kbench.py cfg5 $V/lib_base.so $V/lib_pf1.so > $OUT/kb_cfg5.txt 2>&1; cat $OUT/kb_cfg5.txt
timeout 300 python tools/kbench.py --flags=2 cfg5 $V/lib_base.so $V/lib_pf1.so > $OUT/kb_cfg5_f2.txt 2>&1; cat $OUT/kb_cfg5_f2.txt
timeout 300 python tools/kbench.py --flags=4 cfg5 $V/lib_base.so > $OUT/kb_cfg5_f4.txt 2>&1; cat $OUT/kb_cfg5_f4.txt
timeout 300 python tools/kbench.py cfg3 $V/lib_base.so $V/lib_pf1.so > $OUT/kb_cfg3.txt 2>&1; cat $OUT/kb_cfg3.txt
timeout 300 python tools/kbench.py --flags=6 cfg3 $V/lib_base.so > $OUT/kb_cfg3_f6.txt 2>&1; cat $OUT/kb_cfg3_f6.txt
timeout 200 python tools/kbench.py cfg4 $V/lib_base.so $V/lib_c4one.so > $OUT/kb_cfg4.txt 2>&1; cat $OUT/kb_cfg4.txt
timeout 400 python tools/fit_rate.py > $OUT/fit_rate.txt 2>&1; cat $OUT/fit_rate.txt
for c in cfg2 cfg4; do timeout 300 python bench.py --workload $c --no-cpu-baseline > $OUT/bench_$c.txt 2> $OUT/bench_$c.err; tail -1 $OUT/bench_$c.err; done
