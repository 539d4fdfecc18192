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
