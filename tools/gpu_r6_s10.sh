#!/bin/bash
# round 6 session 10: (x, t) evolution shape two-team kernels against the general kernel they replace; per-team phase clocks of BASELINE config 2's kernel, lock-step and with team 1 five barriers behind
TAG=${1:-r6s10}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
V=gpurun_variants
timeout 600 python tools/kbench.py burgers64 $V/lib_p_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_burgers64.txt 2>&1; tail -n 4 $OUT/kbench_burgers64.txt
timeout 600 python tools/kbench.py heat64 $V/lib_p_base.so pydens_amd/libpinn_hip.so > $OUT/kbench_heat64.txt 2>&1; tail -n 4 $OUT/kbench_heat64.txt
timeout 300 python tools/phases.py $V/lib_ph_base.so cfg2 teams > $OUT/phases_cfg2_lockstep.txt 2>&1; cat $OUT/phases_cfg2_lockstep.txt | tail -n 20
timeout 300 python tools/phases.py $V/lib_ph_sk5.so cfg2 teams > $OUT/phases_cfg2_skew5.txt 2>&1; cat $OUT/phases_cfg2_skew5.txt | tail -n 20
timeout 300 python tools/kbench.py cfg2 $V/lib_ph_base.so $V/lib_ph_sk5.so > $OUT/kbench_cfg2_ph.txt 2>&1; tail -n 4 $OUT/kbench_cfg2_ph.txt
for w in burgers64 heat64 poisson512; do timeout 400 python bench.py --workload $w --no-cpu-baseline --no-strong > $OUT/bench_$w.txt 2> $OUT/bench_$w.err; grep 'bench\] gpu' $OUT/bench_$w.err; done
