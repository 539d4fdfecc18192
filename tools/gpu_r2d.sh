#!/bin/bash
# round 2, GPU session D: store-latency experiments on the WGX tile kernel, cfg4 step overhead breakdown
OUT=/root/repo/gpurun_out/r2d; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 300 python tools/kbench.py cfg5 $V/lib_base.so $V/lib_nt.so $V/lib_gzlate.so $V/lib_ntgz.so > $OUT/kb_cfg5.txt 2>&1; cat $OUT/kb_cfg5.txt
timeout 300 python tools/kbench.py cfg3 $V/lib_base.so $V/lib_nt.so $V/lib_gzlate.so $V/lib_ntgz.so > $OUT/kb_cfg3.txt 2>&1; cat $OUT/kb_cfg3.txt
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg4 -- python /root/repo/bench.py --workload cfg4 --no-cpu-baseline > $OUT/bench_cfg4_rocprof.txt 2>&1; cd /root/repo
find $OUT/trace_cfg4 -name "*kernel_stats.csv" -exec cp {} $OUT/cfg4_kernel_stats.csv \;
cut -c1-150 $OUT/cfg4_kernel_stats.csv | head -8
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "layout_breadth or deep_network or heat3d" > $OUT/pytest_sel.log 2>&1; tail -3 $OUT/pytest_sel.log
