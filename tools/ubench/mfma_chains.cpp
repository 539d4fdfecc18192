// v_mfma_f32_16x16x4_f32 issue rate vs the number of independent accumulator chains (gfx950): how far apart must two
// MFMAs that accumulate into the same tile be for the matrix pipe to stay busy?  One wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_chains.cpp -o mfma_chains && ./mfma_chains
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void __launch_bounds__(256, 1) k(int iters, float* out) {
    f32x4 a[CH];
    for (int c = 0; c < CH; ++c) a[c] = f32x4{0, 0, 0, 0};
    const float x = threadIdx.x * 1e-3f, y = 1.0f + x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 120 / CH; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[c], 0, 0, 0);
    }
    float r = 0;
    for (int c = 0; c < CH; ++c) r += a[c][0] + a[c][3];
    if (r == 12345.678f) out[0] = r;
}

template <int CH>
void run(float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<CH>, dim3(256), dim3(256), 0, 0, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double n = (double)iters * (120 / CH) * CH;
    printf("%d chain(s): %7.3f ms for %.0f MFMAs per wave -> %.1f ns each (%.1f cycles at 2.4 GHz; 32 = pipe saturated)\n", CH, ms, n,
           ms * 1e6 / n, ms * 1e6 / n * 2.4);
}

int main() {
    float* out; hipMalloc(&out, 4);
    run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<5>(out); run<6>(out); run<8>(out);
    return 0;
}
