#!/bin/bash
# registers / spills of every tile and weight-gradient kernel instantiation of widths 64 / 128 / 256 (both sets of full breadth kernels) and 512 on the
# CURRENT sources -> profiles/<round>_kernel_resources.txt. Compile only (no GPU), ~10 minutes on 8 cores.
# usage: bash tools/kernel_resources.sh r06
R=${1:-r06}; cd /root/repo; OUT=profiles/${R}_kernel_resources.txt
for hp in 64 128 256; do (SPILL_MAP_UNITS=allact python tools/spill_map.py $hp > /tmp/kres_$hp.txt 2>&1 &); done
(python tools/spill_map.py 512 > /tmp/kres_512.txt 2>&1 &)          # (width 512: plain kernels + the first breadth set only)
while pgrep -f "tools/spill_map.py" > /dev/null; do sleep 5; done
HASH=$(python -c "from pydens_amd.csrc import build; print(build.kernel_sources_sha1())")
{
cat <<HDR
# registers / spills of every tile and weight-gradient kernel instantiation of widths 64 / 128 / 256 -- ALL THREE widths, both sets of full breadth kernels -- and 512
# on the sources with kernel_sources_sha1 $HASH (tools/kernel_resources.sh -> tools/spill_map.py: hipcc -S per translation unit, .vgpr_spill_count,
# scratch stores / loads by where they sit).
# Columns: spilled VGPRs | MFMAs | "span": scratch instructions BETWEEN the first and the last MFMA of a barrier-delimited segment that holds a GEMM (>= 16 MFMAs), or
# anywhere in a barrier-free loop that holds MFMAs -- what a K step would wait for | "gemm-seg": all scratch instructions of such segments (the span + the operand
# staging / epilogue code that shares the segment with the GEMM: paid once per tile and phase) | "other": segments without a GEMM (first layer, point stage, ...).
# ACTC column (6th template argument of pinn_tile_kernel): -1 = first set of full breadth kernels (activation codes 0 .. 7), -2 = second set (all sixteen, nested
# skips); the last template argument of pinn_wgrad_kernel: true = ALLACT partner, the one before: HEAVY.
# (round 5's file counted every scratch instruction of a GEMM SEGMENT as "inside a GEMM loop"; the tile kernels are fully unrolled and a segment runs from barrier to
#  barrier, i.e. GEMM + the jet epilogue behind it -- the span column separates the two.)
HDR
for hp in 64 128 256 512; do
  echo; cat /tmp/kres_$hp.txt
  echo "# width $hp: kernels with scratch traffic inside an MFMA span: $(grep -c 'span st' /tmp/kres_$hp.txt | tr -d '\n') listed, $(awk '/span st +[1-9]|span st +[0-9]+ ld +[1-9]/' /tmp/kres_$hp.txt | wc -l) with a non-zero span; worst spill count $(awk '{print $0}' /tmp/kres_$hp.txt | grep -o '>  *[0-9]*   mfma' | awk '{print $2}' | sort -n | tail -1)"
done
} > $OUT
tail -3 $OUT
