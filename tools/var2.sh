#!/bin/bash
# Experiment builds behind DESIGN.md's account of the "two workgroups per CU" finding: the BASELINE width-64
# kernels as TWO INDEPENDENT workgroups per CU (VAR 2) instead of two teams in one, exact-fp32 and split-bf16, plain / with every
# wait forced to zero / with a scratch canary. -> gpurun_variants/lib_v2*.so; run with tools/gpu_var2.sh
cd /root/repo
V2="-DPINN_SPLIT_VAR=2 -DPINN_VAR2_REFUSED=0 -DPINN_CFG2_VAR=2 -DPINN_FAST_VAR_S2=(48|2)"
export VARIANT_WIDTHS=64
tools/variant.sh base
tools/variant.sh v2 $V2
tools/variant.sh v2_fz $V2 -mllvm -amdgpu-waitcnt-forcezero
tools/variant.sh v2_canary $V2 -DPINN_SCRATCH_CANARY
tools/variant.sh base_canary -DPINN_SCRATCH_CANARY
# second round: what in the split kernel of config 2 depends on the drift? (operand prefetch forms; every fragment load behind the
# COMPLETION of the MFMAs issued before it = PINN_SP_DRAIN)
P00='{"1": ["-DPINN_SP_PIPE=0", "-DPINN_SP_PIPE_W=0"], "2": ["-fno-slp-vectorize"]}'
P11='{"1": ["-DPINN_SP_PIPE=1", "-DPINN_SP_PIPE_W=1"], "2": ["-fno-slp-vectorize"]}'
PINN_SPLIT_FLAGS=$P11 tools/variant.sh v2_pipe11 $V2
PINN_SPLIT_FLAGS=$P00 tools/variant.sh v2_pipe00 $V2
PINN_SPLIT_FLAGS=$P00 tools/variant.sh v2_il0 $V2 -DPINN_SCHED_IL=0
PINN_SPLIT_FLAGS=$P00 tools/variant.sh v2_drain $V2 -DPINN_SCHED_IL=0 -DPINN_SP_DRAIN=1
