#!/bin/bash
# round 2, GPU session B: parity tests, variant A/B (cfg3/cfg4/cfg5), phases of the WGX tile kernels, first profile pass
OUT=/root/repo/gpurun_out/r2b; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
timeout 300 python tools/kbench.py cfg5 $V/lib_base.so $V/lib_nowgx.so > $OUT/kb_cfg5.txt 2>&1; cat $OUT/kb_cfg5.txt
timeout 300 python tools/kbench.py cfg3 $V/lib_base.so $V/lib_nowgx.so $V/lib_c3mt2.so $V/lib_c3one.so > $OUT/kb_cfg3.txt 2>&1; cat $OUT/kb_cfg3.txt
timeout 200 python tools/kbench.py cfg4 $V/lib_base.so $V/lib_c4w2.so $V/lib_c4w2m1.so > $OUT/kb_cfg4.txt 2>&1; cat $OUT/kb_cfg4.txt
for c in cfg3 cfg5; do timeout 200 python tools/phases.py $V/lib_phases.so $c > $OUT/phases_$c.txt 2>&1; cat $OUT/phases_$c.txt; done
timeout 200 python bench.py --workload cfg4 --no-cpu-baseline --unfused > $OUT/bench_cfg4_unfused.txt 2> $OUT/bench_cfg4_unfused.err; tail -2 $OUT/bench_cfg4_unfused.err; cut -c1-300 $OUT/bench_cfg4_unfused.txt
timeout 600 bash tools/profile_bench.sh cfg5 r2b > /dev/null 2>&1; cat $OUT/prof_cfg5/summary.txt
timeout 600 bash tools/profile_bench.sh cfg3 r2b > /dev/null 2>&1; cat $OUT/prof_cfg3/summary.txt
