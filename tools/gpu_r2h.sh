#!/bin/bash
OUT=/root/repo/gpurun_out/r2h; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 300 python tools/kbench.py cfg5 $V/lib_base.so > $OUT/kb_cfg5.txt 2>&1; cat $OUT/kb_cfg5.txt
timeout 300 python tools/kbench.py cfg3 $V/lib_base.so > $OUT/kb_cfg3.txt 2>&1; cat $OUT/kb_cfg3.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cfg3 or cfg5 or streamed or third" > $OUT/pytest_sel.log 2>&1; tail -2 $OUT/pytest_sel.log
