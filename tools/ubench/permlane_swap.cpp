// Does v_permlane16_swap / v_permlane32_swap (gfx950) give a lane-wise sum over the four 16-lane rows of a wave, and
// what does it cost next to the ds_bpermute form of __shfl_xor? (hipcc 7.2 lowers the permlane*_swap BUILTINS to code
// that adds the first result to itself, so the kernels use the instruction through inline asm.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/permlane_swap.cpp -o permlane_swap && ./permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float rows_sum_swap(float x) {
    float y = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    x += y;
    y = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float rows_sum_shfl(float x) {
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}
template <int MODE>
__global__ void k(const float* in, float* out, int reps) {
    float v = in[threadIdx.x], acc = 0.0f;
    for (int i = 0; i < reps; ++i) {
        const float s = MODE ? rows_sum_swap(v) : rows_sum_shfl(v);
        acc += s;
        v = v * 1.0001f + 0.5f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const int n = 64, reps = 4096, blocks = 1024;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 0.25f * i - 3.0f;
    float *in, *o0, *o1;
    hipMalloc(&in, n * 4); hipMalloc(&o0, blocks * n * 4); hipMalloc(&o1, blocks * n * 4);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int mode = 0; mode < 2; ++mode) {
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0);
            if (mode) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(n), 0, 0, in, o1, reps);
            else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(n), 0, 0, in, o0, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mode], e0, e1);
        }
    }
    std::vector<float> a(n), b(n);
    hipMemcpy(a.data(), o0, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), o1, n * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < n; ++i) worst = fmax(worst, fabs((double)a[i] - b[i]) / fmax(1.0, fabs((double)a[i])));
    printf("rows-sum via __shfl_xor: %.3f ms   via permlane16/32_swap: %.3f ms   max rel difference %.3g\n", ms[0], ms[1], worst);
    return worst < 1e-6 ? 0 : 1;
}
