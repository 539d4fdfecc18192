// Which instruction classes overlap with v_mfma_f32_16x16x4_f32 on gfx950 -- from ANOTHER wave on the same SIMD, and
// from the SAME wave (issued between the MFMAs)? Classes: fp32 FMA, packed fp32 FMA, transcendental (v_exp_f32),
// integer add, v_mov, LDS read (ds_read_b128).
//   time(both) ~ max(solo times)  -> the class runs in the shadow of the matrix pipe
//   time(both) ~ sum(solo times)  -> it shares the datapath (or the issue slot) with the MFMA
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_coexec_matrix.cpp -o mfma_coexec_matrix && ./mfma_coexec_matrix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int CLS>
__device__ __forceinline__ void other8(float (&v)[8], f32x2 (&p)[8], int (&n)[8], f32x4 (&l)[8], const float* lds) {
    if (CLS == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(1.0001f), "v"(0.5f));
        REP8(X)
#undef X
    } else if (CLS == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f32x2{1.0001f, 1.0001f}), "v"(f32x2{0.5f, 0.5f}));
        REP8(X)
#undef X
    } else if (CLS == 2) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        REP8(X)
#undef X
    } else if (CLS == 3) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(3));
        REP8(X)
#undef X
    } else if (CLS == 4) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(v[(i + 1) & 7]));
        REP8(X)
#undef X
    } else {
#define X(i) l[i] = *reinterpret_cast<const f32x4*>(lds + ((threadIdx.x * 4 + i * 1024) & 8191));
        REP8(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// MODE 0: MFMA waves only (waves 0-3); 1: waves 4-7 only, the other class; 2: one wave of each kind per SIMD;
// 3: ONE wave per SIMD issuing 4 MFMAs then 8 ops of the class, repeated; 4: that wave's MFMAs alone; 5: its ops alone.
// (MODE is a template parameter: no branches inside the timed loops.)
template <int CLS, int MODE>
__global__ void __launch_bounds__(512, 2) k(int iters, float* out) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float v[8]; f32x2 p[8]; int n[8]; f32x4 l[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 1e-3f + i; p[i] = f32x2{v[i], v[i]}; n[i] = threadIdx.x + i; l[i] = f32x4{0, 0, 0, 0}; }
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = threadIdx.x * 1e-3f, y = 1.0f + x;
    constexpr bool SAME = MODE >= 3;
    const bool mf = SAME ? (wave < 4 && MODE != 5) : (wave < 4 && MODE != 1);
    const bool ot = SAME ? (wave < 4 && MODE != 4) : (wave >= 4 && MODE != 0);
    if (mf && ot) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
                other8<CLS>(v, p, n, l, lds);
            }
        }
    } else if (mf) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            }
        }
    } else if (ot) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) other8<CLS>(v, p, n, l, lds);
        }
    }
    float r = a0[0] + a1[1] + a2[2] + a3[3];
    for (int i = 0; i < 8; ++i) r += v[i] + p[i][0] + p[i][1] + n[i] + l[i][0];
    if (r == 12345.678f) out[0] = r;
}

template <int CLS, int MODE>
float time_mode(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<CLS, MODE>), dim3(256), dim3(512), 0, 0, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

template <int CLS>
void run(const char* name, float* out) {
    const int iters = 20000;
    const float ms[6] = {time_mode<CLS, 0>(out, iters), time_mode<CLS, 1>(out, iters), time_mode<CLS, 2>(out, iters),
                         time_mode<CLS, 3>(out, iters), time_mode<CLS, 4>(out, iters), time_mode<CLS, 5>(out, iters)};
    auto verdict = [](float a, float b, float both) { return both < 0.5f * (a + b) + 0.5f * fmaxf(a, b) ? "overlaps" : "serialises"; };
    printf("%-14s other wave: mfma %.2f  other %.2f  both %.2f ms (%s) | same wave: mfma %.2f  other %.2f  both %.2f ms (%s)\n", name,
           ms[0], ms[1], ms[2], verdict(ms[0], ms[1], ms[2]), ms[4], ms[5], ms[3], verdict(ms[4], ms[5], ms[3]));
}

int main() {
    float* out; hipMalloc(&out, 4);
    printf("per step: 4 MFMAs (128 cycles of matrix pipe) and / or 8 instructions of a class; 80000 steps, 256 CUs\n");
    run<0>("v_fma_f32", out);
    run<1>("v_pk_fma_f32", out);
    run<2>("v_exp_f32", out);
    run<3>("v_add_u32", out);
    run<4>("v_mov_b32", out);
    run<5>("ds_read_b128", out);
    return 0;
}
