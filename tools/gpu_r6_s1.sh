#!/bin/bash
# round 6 session 1: same-box A/B of the team-skew / top-barrier variants (tools/variant.sh builds) on BASELINE configs 2 and 4,
# then the probe of config 4's output-bias gradient. usage: gpurun -- 'bash tools/gpu_r6_s1.sh <tag>'
TAG=${1:-r6s1}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
LIBS="$V/lib_r6_base.so $V/lib_r6_notopb.so $V/lib_r6_nt_sk3.so $V/lib_r6_nt_sk5.so $V/lib_r6_nt_sk7.so $V/lib_r6_sk6.so"
timeout 600 python tools/kbench.py cfg2 $LIBS > $OUT/kbench_cfg2.txt 2>&1
timeout 600 python tools/kbench.py cfg4 $LIBS > $OUT/kbench_cfg4.txt 2>&1
timeout 600 python tools/cfg4_bl_probe.py 99 100 101 > $OUT/cfg4_bl_probe.txt 2>&1
tail -n 30 $OUT/kbench_cfg2.txt $OUT/kbench_cfg4.txt
tail -n 60 $OUT/cfg4_bl_probe.txt
