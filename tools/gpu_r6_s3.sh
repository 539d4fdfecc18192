#!/bin/bash
# round 6 session 3: the whole GPU tier on the fp64 pre-pass / fp64 row-sum sources + the config-4 probe + kernel-time A/B against the build before
TAG=${1:-r6s3}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 600 python tools/cfg4_bl_probe.py 99 100 101 > $OUT/cfg4_bl_probe.txt 2>&1; grep -A3 "^---" $OUT/cfg4_bl_probe.txt | head -24
timeout 600 python tools/kbench.py cfg2 gpurun_variants/lib_r6_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_cfg2.txt 2>&1; tail -4 $OUT/kbench_cfg2.txt
timeout 600 python tools/kbench.py cfg4 gpurun_variants/lib_r6_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_cfg4.txt 2>&1; tail -4 $OUT/kbench_cfg4.txt
timeout 3000 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt
cp gpurun_out/grad_margins.txt $OUT/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
