// emu_runtime.h -- TEST INFRASTRUCTURE ONLY.  Host-side SIMT emulator for the port layer of
// pydens_amd/csrc/pinn_port.h: every workgroup thread is a ucontext fiber, scheduled round-robin on one OS
// thread; __syncthreads() and the wave-collective primitives (MFMA, shuffles) are rendezvous points.
// The emulated v_mfma_f32_16x16x4_f32 follows the lane maps documented for gfx950
// (/opt/skills/guides/cdna_hip_programming.md section 3): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// D row = (l>>4)*4 + reg, col = l&15, accumulated as a k-ordered fmaf chain.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct float4 { float x, y, z, w; };

namespace emu {
int tid();
int bid();
int nblk();
float* smem();
void sync_block();
void sync_wave();
void yield_fiber();
f32x4 mfma16(float a, float b, f32x4 c);
float shfl_xor(float v, int mask);
f32x4 mfma16_bf16(const unsigned short* a8, const unsigned short* b8, f32x4 c);
void lds_tr16(const void* p, unsigned short* out4);
float row_sum16(float v);
void launch(int grid, int block, size_t smem_bytes, const std::function<void()>& body);
}

#define PINN_GLOBAL
#define PINN_DEVICE static inline
#define PINN_TID (emu::tid())
#define PINN_BID (emu::bid())
#define PINN_NBLK (emu::nblk())
#define PINN_SYNC() emu::sync_block()
#define PINN_FENCE_BLOCK()
#define PINN_WAVE_SYNC() emu::sync_wave()
#define PINN_SMEM(name) float* name = emu::smem()
#define PINN_LAUNCH_BOUNDS(n)

static inline f32x4 pinn_mfma16(float a, float b, f32x4 c) { return emu::mfma16(a, b, c); }
typedef short pinn_s16x8 __attribute__((ext_vector_type(8)));
typedef short pinn_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int pinn_u32x2 __attribute__((ext_vector_type(2)));
// v_mfma_f32_16x16x32_bf16: products of bf16 values are exact; the sum over the 32 k slots and the accumulator is taken in
// double here and rounded once (the hardware's internal order is not documented; tools/ubench/split_bf16.cpp measures it
// at or below the error of an fp32 fmaf chain)
static inline f32x4 pinn_mfma16_bf16(pinn_s16x8 a, pinn_s16x8 b, f32x4 c) {
    unsigned short aa[8], bb[8];
    for (int e = 0; e < 8; ++e) { aa[e] = (unsigned short)a[e]; bb[e] = (unsigned short)b[e]; }
    return emu::mfma16_bf16(aa, bb, c);
}
static inline pinn_s16x4 pinn_lds_tr16(const void* p) {
    unsigned short o[4];
    emu::lds_tr16(p, o);
    return pinn_s16x4{(short)o[0], (short)o[1], (short)o[2], (short)o[3]};
}
static inline unsigned pinn_pack_hi16(unsigned a, unsigned b) { return (a >> 16) | (b & 0xffff0000u); }
static inline int pinn_flag_load(const int* p) { return *reinterpret_cast<const volatile int*>(p); }
static inline void pinn_flag_publish(int* p, int v, bool leader) {
    emu::sync_wave();                       // every lane's data is in place before the leader raises the flag
    if (leader) *reinterpret_cast<volatile int*>(p) = v;
    emu::sync_wave();
}
#define PINN_SPIN_PAUSE() emu::yield_fiber()
static inline void pinn_flag_arrive(int* p, bool leader) {
    emu::sync_wave();
    if (leader) *reinterpret_cast<volatile int*>(p) += 1;
    emu::sync_wave();
}
struct PinnRows { char* p; };
static inline PinnRows pinn_rows(const void* base, unsigned) { return PinnRows{(char*)const_cast<void*>(base)}; }
static inline f32x4 pinn_rows_ld4(const PinnRows& b, int lane_bytes, int row_bytes) { return *reinterpret_cast<const f32x4*>(b.p + lane_bytes + row_bytes); }
static inline void pinn_rows_st4(const PinnRows& b, int lane_bytes, int row_bytes, f32x4 v) { *reinterpret_cast<f32x4*>(b.p + lane_bytes + row_bytes) = v; }
static inline int pinn_wave_uniform(int v) { return v; }
static inline float pinn_shfl_xor(float v, int mask) { return emu::shfl_xor(v, mask); }
static inline float pinn_rows_sum(float x) {
    x += emu::shfl_xor(x, 16);
    x += emu::shfl_xor(x, 32);
    return x;
}
static inline float pinn_row_sum16(float v) { return emu::row_sum16(v); }
// (doubles across lanes: two 32-bit exchanges per step, like the device's DPP / readlane forms in pinn_port.h)
static inline double pinn_emu_shfl_xor_f64(double v, int mask) {
    unsigned long long b; memcpy(&b, &v, 8);
    unsigned lo = (unsigned)(b & 0xffffffffull), hi = (unsigned)(b >> 32);
    float flo, fhi; memcpy(&flo, &lo, 4); memcpy(&fhi, &hi, 4);
    flo = emu::shfl_xor(flo, mask); fhi = emu::shfl_xor(fhi, mask);
    memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
    b = ((unsigned long long)hi << 32) | lo;
    double r; memcpy(&r, &b, 8); return r;
}
static inline double pinn_row_sum16_f64(double v) { for (int m = 1; m < 16; m <<= 1) v += pinn_emu_shfl_xor_f64(v, m); return v; }
static inline double pinn_rows_total_f64(double v) { v += pinn_emu_shfl_xor_f64(v, 16); v += pinn_emu_shfl_xor_f64(v, 32); return v; }
template <int N> static inline void pinn_row_sum16_n(float (&v)[N]) {
    for (int i = 0; i < N; ++i) v[i] = emu::row_sum16(v[i]);
}
static inline float pinn_exp2(float x) { return exp2f(x); }
static inline float pinn_rcp(float x) { return 1.0f / x; }
#define PINN_LAUNCH_BOUNDS2(n, w)
#define PINN_SCHED_BARRIER()
#define PINN_SCHED_IL 0
template <int N_MFMA, int N_MEM> static inline void pinn_sched_interleave() {}
template <int N_DS, int N_MFMA> static inline void pinn_sched_reads_first() {}
#define PINN_INLINE_LAMBDA
#define PINN_SETPRIO(n)
