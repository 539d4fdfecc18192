#!/bin/bash
# rocprofv3 evidence for ONE bench.py workload (the same command the bench line comes from): kernel trace + stats, then
# PMC passes, each in its own run (counters never share a run with a trace domain other than --kernel-trace).
# (--no-parity --no-cold: every launch of a matrix kernel in the profile has the workload's full grid and runs in the settled clock
#  state, so that rocprofv3's own AverageNs is the duration the bench line prices -- VERDICT r4 item 1)
# usage: tools/profile_bench.sh <cfg2|cfg3|cfg4|cfg5> <tag> [suffix] [extra bench.py arguments, e.g. --gemm bf16x3]
#        -> gpurun_out/<tag>/prof_<cfg><suffix>/, summary + JSON for profiles/
CFG=${1:-cfg2}; TAG=${2:-prof}; SUF=${3:-}; shift 3 2>/dev/null; EXTRA="$@"; OUT=/root/repo/gpurun_out/$TAG/prof_$CFG$SUF; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
STEPS=30; WARM=5; case $CFG in cfg3|cfg5) STEPS=24; WARM=4;; esac    # (wide configs: enough launches that the first-touch launches of the 12 GB slabs do not weigh on AverageNs)
CMD="python /root/repo/bench.py --workload $CFG --no-cpu-baseline --no-strong --no-side --no-parity --no-cold --no-configs --steps $STEPS --warmup $WARM $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc3 -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc4 -- $CMD > $OUT/pmc4.log 2>&1
cd /root/repo
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python tools/summarize_bench_prof.py $CFG $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
cat $OUT/summary.txt
