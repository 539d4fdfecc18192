// pinn_port.h -- the handful of execution-model primitives the kernels use.
// Product build (hipcc, gfx950): thin names over HIP builtins.  Test build (-DPINN_EMU, host clang++):
// tests/emu/emu_runtime.h provides a fiber-based SIMT emulator with the same names, so that the very same
// kernel source (indexing, barriers, MFMA operand/accumulator lane maps) can be checked against the oracle
// on a machine without a GPU.  The emulator is test infrastructure: libpinn_hip.so never contains it.
#pragma once

#if defined(PINN_EMU)
#include "emu_runtime.h"
#else
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PINN_GLOBAL __global__
#define PINN_DEVICE __device__ __forceinline__
#define PINN_TID ((int)threadIdx.x)
#define PINN_BID ((int)blockIdx.x)
#define PINN_NBLK ((int)gridDim.x)
#define PINN_SYNC() __syncthreads()
#define PINN_SMEM(name) extern __shared__ __attribute__((aligned(16))) float name[]
#define PINN_LAUNCH_BOUNDS(n) __launch_bounds__(n)

// D[16x16] += A[16x4] * B[4x16], exact fp32 (v_mfma_f32_16x16x4_f32).
// lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; c[r] is D[(l>>4)*4 + r][l&15].
PINN_DEVICE f32x4 pinn_mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
PINN_DEVICE float pinn_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
#endif
