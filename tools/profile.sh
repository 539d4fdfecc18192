#!/bin/bash
# rocprofv3 passes for one BASELINE config: kernel trace + stats, then PMC passes (each in its own run).
# usage: tools/profile.sh <cfg> <outdir-under-gpurun_out>
CFG=${1:-cfg2}; OUT=/root/repo/gpurun_out/${2:-prof}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/tools/prof_run.py $CFG 20 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc1 -- python /root/repo/tools/prof_run.py $CFG 5 > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU -d $OUT/pmc2 -- python /root/repo/tools/prof_run.py $CFG 5 > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -- python /root/repo/tools/prof_run.py $CFG 5 > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc4 -- python /root/repo/tools/prof_run.py $CFG 5 > $OUT/pmc4.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_COEXEC_CYCLES -d $OUT/pmc5 -- python /root/repo/tools/prof_run.py $CFG 5 > $OUT/pmc5.log 2>&1
cd /root/repo; python tools/summarize_prof.py $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
