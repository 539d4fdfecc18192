#!/bin/bash
# round 2, GPU session A: parity tests, A/B of the kernel variants (gpurun_variants/), tanh arbiter, phase profiles, bench smoke
OUT=/root/repo/gpurun_out/r2a; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
timeout 300 python tools/kbench.py cfg5 $V/lib_base.so $V/lib_nowgx.so > $OUT/kb_cfg5.txt 2>&1; cat $OUT/kb_cfg5.txt
timeout 300 python tools/kbench.py cfg3 $V/lib_base.so $V/lib_nowgx.so > $OUT/kb_cfg3.txt 2>&1; cat $OUT/kb_cfg3.txt
timeout 200 python tools/kbench.py cfg4 $V/lib_base.so $V/lib_mt4.so $V/lib_tanh1.so $V/lib_tanh5.so > $OUT/kb_cfg4.txt 2>&1; cat $OUT/kb_cfg4.txt
timeout 200 python tools/kbench.py cfg2 $V/lib_base.so $V/lib_tanh1.so $V/lib_tanh5.so > $OUT/kb_cfg2.txt 2>&1; cat $OUT/kb_cfg2.txt
timeout 400 python tools/arbiter.py $V/lib_base.so $V/lib_tanh1.so $V/lib_tanh5.so cfg2 cfg4 cfg3 > $OUT/arbiter.txt 2>&1; cat $OUT/arbiter.txt
for c in cfg4 cfg5 cfg3 cfg2; do timeout 200 python tools/phases.py $V/lib_phases.so $c > $OUT/phases_$c.txt 2>&1; cat $OUT/phases_$c.txt; done
for c in cfg2 cfg3 cfg4 cfg5; do timeout 300 python bench.py --workload $c --no-cpu-baseline > $OUT/bench_$c.txt 2> $OUT/bench_$c.err; tail -1 $OUT/bench_$c.err; cut -c1-600 $OUT/bench_$c.txt; done
timeout 200 python bench.py --workload cfg4 --no-cpu-baseline --unfused > $OUT/bench_cfg4_unfused.txt 2> $OUT/bench_cfg4_unfused.err; tail -2 $OUT/bench_cfg4_unfused.err; cut -c1-300 $OUT/bench_cfg4_unfused.txt
