#!/bin/bash
OUT=/root/repo/gpurun_out/r2k; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 300 python tools/kbench.py cfg2 $V/lib_base.so $V/lib_c2old.so > $OUT/kb_cfg2.txt 2>&1; cat $OUT/kb_cfg2.txt
timeout 300 python tools/kbench.py cfg4 $V/lib_base.so $V/lib_c4w2.so > $OUT/kb_cfg4.txt 2>&1; cat $OUT/kb_cfg4.txt
for l in base c2old base c2old; do timeout 200 python bench.py --workload cfg2 --no-cpu-baseline --lib $V/lib_$l.so 2>&1 >/dev/null | grep "gpu:" | sed "s/^/$l cfg2 /"; done | tee $OUT/step_cfg2.txt
for l in base c4w2 base c4w2; do timeout 200 python bench.py --workload cfg4 --no-cpu-baseline --lib $V/lib_$l.so 2>&1 >/dev/null | grep "gpu:" | sed "s/^/$l cfg4 /"; done | tee $OUT/step_cfg4.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cfg2 or cfg4 or ragged or golden or sharded or tutorial or known" > $OUT/pytest_sel.log 2>&1; tail -3 $OUT/pytest_sel.log
