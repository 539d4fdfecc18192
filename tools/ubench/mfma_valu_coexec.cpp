// Micro-benchmark: does v_mfma_f32_16x16x4_f32 co-execute with fp32 VALU work of ANOTHER wave on the same SIMD (gfx950)?
// One workgroup of 512 threads per CU = 2 waves per SIMD. Mode 0: both waves MFMA; 1: both VALU; 2: waves 0-3 MFMA and
// waves 4-7 VALU (one of each per SIMD); 3: only waves 0-3 MFMA; 4: only waves 4-7 VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512, 2) k(int mode, int iters, float* out) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = (mode == 0) || ((mode == 2 || mode == 3) && wave < 4);
    const bool do_valu = (mode == 1) || ((mode == 2 || mode == 4) && wave >= 4);
    float r = 0.f;
    if (do_mfma) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0;
        const float x = threadIdx.x * 1e-3f, y = 1.0f + x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
                a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1] + a2[2] + a3[3] + a4[0];
    } else if (do_valu) {
        float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
        const float c = 1.0001f, e = 0.5f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 10; ++u) {      // 80 dependent-free v_fma_f32 per iteration (8 chains)
                v0 = __builtin_fmaf(v0, c, e); v1 = __builtin_fmaf(v1, c, e); v2 = __builtin_fmaf(v2, c, e); v3 = __builtin_fmaf(v3, c, e);
                v4 = __builtin_fmaf(v4, c, e); v5 = __builtin_fmaf(v5, c, e); v6 = __builtin_fmaf(v6, c, e); v7 = __builtin_fmaf(v7, c, e);
            }
        }
        r = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
    if (r == 12345.678f) out[0] = r;
}

int main() {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[5] = {"both MFMA (2 waves/SIMD)", "both VALU (2 waves/SIMD)", "MFMA wave + VALU wave per SIMD", "MFMA waves only", "VALU waves only"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 5; ++mode) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per SIMD: MFMA wave issues iters*20 MFMAs (32 cycles each); VALU wave iters*80 FMAs
            if (rep) printf("mode %d %-34s %8.3f ms  (MFMA %d x 32 cyc = %.2f Mcyc, VALU %d instr)\n", mode, names[mode], ms,
                            iters * 20, iters * 20 * 32 / 1e6, iters * 80);
        }
    return 0;
}
