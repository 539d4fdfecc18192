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
#define PINN_FENCE_BLOCK() __threadfence_block()
// LDS exchange between the lanes of ONE wave: LDS instructions of a wave execute in issue order, so nothing has to be
// waited for -- the fences only pin the program order of the stores in front and the loads behind
#define PINN_WAVE_SYNC()                                          \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)
#define PINN_SMEM(name) extern __shared__ __attribute__((aligned(16))) float name[]
#define PINN_LAUNCH_BOUNDS(n) __launch_bounds__(n)

// D[16x16] += A[16x4] * B[4x16], exact fp32 (v_mfma_f32_16x16x4_f32).
// lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; c[r] is D[(l>>4)*4 + r][l&15].
PINN_DEVICE f32x4 pinn_mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// ---- split-bf16 GEMM operands (round 3): an fp32 value is EXACTLY hi + mid + lo with three bf16 (8 + 8 + 8 mantissa bits), and
// v_mfma_f32_16x16x32_bf16 multiplies such parts exactly and accumulates in fp32 -- on the matrix pipe proper (16 cycles per
// K = 32 against 8 x 32 cycles of v_mfma_f32_16x16x4_f32, which runs on the fp32 vector lanes) and beside the VALU, not instead of it
// (tools/ubench/split_bf16.cpp: accuracy, issue rates, the transpose read).
typedef short pinn_s16x8 __attribute__((ext_vector_type(8)));   // 8 bf16 bit patterns = 4 VGPRs: one MFMA operand fragment
typedef short pinn_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int pinn_u32x2 __attribute__((ext_vector_type(2)));
// D[16x16] += A[16x32] * B[32x16]: lane l supplies A[l&15][8*(l>>4) + e] and B[8*(l>>4) + e][l&15], e = 0..7 (element e in
// bits 16*(e&1) of register e>>1); c[r] is D[(l>>4)*4 + r][l&15] as for pinn_mfma16.
PINN_DEVICE f32x4 pinn_mfma16_bf16(pinn_s16x8 a, pinn_s16x8 b, f32x4 c) {
    typedef __bf16 pinn_bf16x8 __attribute__((ext_vector_type(8)));
#if defined(PINN_ABL) && (PINN_ABL & 64)          // timing ablation (experiment builds): operands consumed, nothing multiplied
    asm volatile("" :: "v"(a), "v"(b));
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pinn_bf16x8, a), __builtin_bit_cast(pinn_bf16x8, b), c, 0, 0, 0);
}
// ds_read_b64_tr_b16: within each group of 16 lanes, lane i supplies the (8-byte aligned) address of 4 consecutive 16-bit
// elements -- read them as row i/4, columns 4*(i%4) .. +3 of a [4][16] block whose row stride is whatever the addresses say --
// and lane n receives column n of the block: elements (row 0..3, column n). Checked on the device by tools/ubench/split_bf16.cpp.
PINN_DEVICE pinn_s16x4 pinn_lds_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) pinn_s16x4*)(p));
}
// (a >> 16) | (b & 0xffff0000): the bf16 truncations of two fp32 bit patterns in one register (v_perm_b32)
PINN_DEVICE unsigned pinn_pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
// flags between two waves of a workgroup in LDS (a producer / consumer pair): LDS instructions of a CU execute in issue order, so
// "data, then flag" on one side and "flag, then data" on the other need no waiting -- only the program order, which the
// wavefront-scope fences pin (a workgroup-scope release would wait for every outstanding GLOBAL access of the wave as well)
PINN_DEVICE int pinn_flag_load(const int* p) { return *reinterpret_cast<const volatile int*>(p); }
PINN_DEVICE void pinn_flag_publish(int* p, int v, bool leader) {
    PINN_WAVE_SYNC();
    if (leader) *reinterpret_cast<volatile int*>(p) = v;
    PINN_WAVE_SYNC();
}
#define PINN_SPIN_PAUSE() __builtin_amdgcn_s_sleep(1)
// arrival counter of a barrier among SOME waves of a workgroup (a team), in LDS: every wave adds one after its own LDS accesses
// (program order + in-order LDS = the same guarantee s_barrier gives for LDS data) and polls for the round's total
PINN_DEVICE void pinn_flag_arrive(int* p, bool leader) {
    PINN_WAVE_SYNC();
    if (leader) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    PINN_WAVE_SYNC();
}
// lane-private f32x4 rows of a wave-owned block of global memory through buffer instructions: ONE vector register of offsets
// (lane * 16) for every row, the row offset in a scalar register -- with plain pointers hipcc keeps a 64-bit address pair
// per row alive across the whole tile loop (several dozen VGPRs, spilled)
struct PinnRows { __amdgpu_buffer_rsrc_t r; };
PINN_DEVICE PinnRows pinn_rows(const void* base /*wave-uniform*/, unsigned bytes) {
    return PinnRows{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000)};
}
typedef unsigned int pinn_u32x4 __attribute__((ext_vector_type(4)));
PINN_DEVICE f32x4 pinn_rows_ld4(const PinnRows& b, int lane_bytes, int row_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, lane_bytes, row_bytes, 0));
}
PINN_DEVICE void pinn_rows_st4(const PinnRows& b, int lane_bytes, int row_bytes, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pinn_u32x4, v), b.r, lane_bytes, row_bytes, 0);
}
// a value the whole wave agrees on, moved to a scalar register (addresses built from it use the scalar-base forms)
PINN_DEVICE int pinn_wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
PINN_DEVICE float pinn_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
// lane-wise sum over the four 16-lane rows of the wave (every lane gets x[l] + x[l^16] + x[l^32] + x[l^48]) with the
// gfx950 row swaps: v_permlane16_swap exchanges the odd rows of its first operand with the even rows of the second,
// v_permlane32_swap the upper half of the first with the lower half of the second, so (a', b') = swap(x, x) holds x[l]
// and x[l ^ 16] (resp. x[l ^ 32]) in every lane. Two VALU ops per step instead of a ds_bpermute round trip. Inline asm:
// hipcc 7.2 lowers the __builtin_amdgcn_permlane*_swap builtins to code that adds the FIRST result to itself
// (tools/ubench/permlane_swap.cpp checks the asm form against __shfl_xor on the device).
PINN_DEVICE float pinn_rows_sum(float x) {
    float y = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    x += y;
    y = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}

// sum over the 16 lanes of a DPP row (lanes sharing lane>>4); every lane of the row gets the sum.
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: four full-rate VALU adds, no LDS.
PINN_DEVICE float pinn_row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}
// ... of a DOUBLE (the head scalars' end-of-workgroup sums, round 6): the two halves through the same DPP moves -- no LDS round trip
PINN_DEVICE double pinn_dpp_f64(double v, const int ctrl) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    int lo = (int)(unsigned)(b & 0xffffffffull), hi = (int)(unsigned)(b >> 32);
    switch (ctrl) {     // (the control word must be an immediate)
        case 0: lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true); break;
        case 1: lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true); break;
        case 2: lo = __builtin_amdgcn_mov_dpp(lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x141, 0xF, 0xF, true); break;
        default: lo = __builtin_amdgcn_mov_dpp(lo, 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x140, 0xF, 0xF, true); break;
    }
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
PINN_DEVICE double pinn_row_sum16_f64(double v) {
    v += pinn_dpp_f64(v, 0); v += pinn_dpp_f64(v, 1); v += pinn_dpp_f64(v, 2); v += pinn_dpp_f64(v, 3);
    return v;
}
// v equal within each 16-lane row -> the sum over the four rows, in every lane (v_readlane: wave-uniform values, no LDS)
PINN_DEVICE double pinn_rows_total_f64(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = (int)(unsigned)(b & 0xffffffffull), hi = (int)(unsigned)(b >> 32);
    double t = 0.0;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const unsigned l = (unsigned)__builtin_amdgcn_readlane(lo, 16 * row), h = (unsigned)__builtin_amdgcn_readlane(hi, 16 * row);
        t += __builtin_bit_cast(double, ((unsigned long long)h << 32) | (unsigned long long)l);
    }
    return t;
}
// the same for N values at once, step-major: N independent adds per DPP step (a lone chain pays two wait states between its
// dependent DPP steps)
template <int N>
PINN_DEVICE void pinn_row_sum16_n(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v[i]), 0xB1, 0xF, 0xF, true));
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v[i]), 0x4E, 0xF, 0xF, true));
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v[i]), 0x141, 0xF, 0xF, true));
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v[i]), 0x140, 0xF, 0xF, true));
}
// 2^x and 1/x at hardware precision (v_exp_f32 / v_rcp_f32, ~1 ulp)
PINN_DEVICE float pinn_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
PINN_DEVICE float pinn_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#define PINN_LAUNCH_BOUNDS2(n, w) __launch_bounds__(n, w)
#define PINN_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// Interleaved issue order inside one scheduling region (between two PINN_SCHED_BARRIERs): N_MEM memory reads (LDS
// and/or global: the operands of the NEXT pipeline stage) spread evenly between the N_MFMA matrix instructions of the
// current stage. A wave issues in order; reads bunched in front of the MFMAs leave the matrix pipe idle while they issue
// (about 70 cycles per 512-cycle batch in the weight-gradient loop), reads placed between two MFMAs issue in the shadow
// of the 32-cycle MFMA before them.
#ifndef PINN_SCHED_IL
#define PINN_SCHED_IL 1
#endif
template <int N_MFMA, int N_MEM>
PINN_DEVICE void pinn_sched_interleave() {
#if PINN_SCHED_IL
    constexpr int PER = (N_MEM > 0) ? (N_MFMA / N_MEM > 0 ? N_MFMA / N_MEM : 1) : N_MFMA;
    constexpr int USED = (N_MEM > 0) ? (PER * N_MEM < N_MFMA ? PER * N_MEM : N_MFMA) : 0;
#pragma unroll
    for (int i = 0; i < N_MEM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);      // MFMA
        __builtin_amdgcn_sched_group_barrier(0x120, 1, 0);        // one DS read or VMEM read
    }
    if (N_MFMA - USED > 0) __builtin_amdgcn_sched_group_barrier(0x008, N_MFMA - USED > 0 ? N_MFMA - USED : 1, 0);
#endif
}
// issue order inside one scheduling region: the next N_DS LDS reads FIRST, then N_MFMA matrix instructions (a fragment
// prefetch must start its round trip before the MFMAs it hides behind; left alone the scheduler sinks it to the end of the
// region, right in front of its first use)
template <int N_DS, int N_MFMA>
PINN_DEVICE void pinn_sched_reads_first() {
    __builtin_amdgcn_sched_group_barrier(0x100, N_DS, 0);        // DS read
    __builtin_amdgcn_sched_group_barrier(0x008, N_MFMA, 0);      // MFMA
}
#define PINN_INLINE_LAMBDA __attribute__((always_inline))
#define PINN_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
