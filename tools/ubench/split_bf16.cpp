// Split-bf16 MFMA feasibility on gfx950 (round 3): can the hidden GEMMs of the tile kernels run on v_mfma_f32_16x16x32_bf16 with
// every fp32 operand split EXACTLY into three bf16 (x = hi + mid + lo), products accumulated in fp32?
//   A  accuracy of C = A B (16 x 64 x 16, one wave) against fp64: exact-fp32 MFMA, 6 products (i + j <= 2), 9 products, two
//      split forms (truncation by masks, round-to-nearest by v_cvt_pk_bf16_f32), two accumulation orders
//   B  ds_read_b64_tr_b16: what each lane receives (the weight-gradient GEMM needs its operands point-contiguous)
//   C  issue: cycles per bf16 MFMA with k = 0..6 single-issue vector instructions of a class placed behind each of them,
//      one and two waves per SIMD (the fp32 MFMA shares the vector issue port, tools/ubench/mfma_coexec_matrix.cpp;
//      does the bf16 one co-execute?)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/split_bf16.cpp -o /tmp/split_bf16 && /tmp/split_bf16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bitsf(unsigned x) { return __builtin_bit_cast(float, x); }

// x = p[0] + p[1] + p[2] exactly, every p a bf16 (returned as floats with 16 low zero bits)
template <int FORM>
__device__ __forceinline__ void split3(float x, float (&p)[3]) {
    if (FORM == 0) {                       // truncation: masks and exact subtractions
        p[0] = bitsf(fbits(x) & 0xffff0000u);
        const float r1 = x - p[0];
        p[1] = bitsf(fbits(r1) & 0xffff0000u);
        const float r2 = r1 - p[1];
        p[2] = bitsf(fbits(r2) & 0xffff0000u);
    } else {                               // round to nearest even (v_cvt_pk_bf16_f32)
        p[0] = (float)(__bf16)x;
        const float r1 = x - p[0];
        p[1] = (float)(__bf16)r1;
        const float r2 = r1 - p[1];
        p[2] = (float)(__bf16)r2;
    }
}
__device__ __forceinline__ __bf16 as_bf16(float hi16) { return __builtin_bit_cast(__bf16, (unsigned short)(fbits(hi16) >> 16)); }

// A [16][64], B [64][16] row-major fp32; out[v][16][16]: v = 0 fp32 MFMA, 1..: split variants
template <int FORM, int NPROD, int ORDER>
__device__ void gemm_split(const float* A, const float* B, float* C) {
    const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    f32x4 acc = {0, 0, 0, 0};
    bf16x8 a[2][3], b[2][3];
    for (int kb = 0; kb < 2; ++kb)
        for (int e = 0; e < 8; ++e) {
            float pa[3], pb[3];
            split3<FORM>(A[lr * 64 + kb * 32 + lq * 8 + e], pa);
            split3<FORM>(B[(kb * 32 + lq * 8 + e) * 16 + lr], pb);
            for (int i = 0; i < 3; ++i) { a[kb][i][e] = as_bf16(pa[i]); b[kb][i][e] = as_bf16(pb[i]); }
        }
    // product list, small terms first (ORDER 0) or last (ORDER 1)
    const int pi[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0}, pj[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0};
    for (int t = 0; t < 9; ++t) {
        const int u = ORDER == 0 ? t : 8 - t;
        if (pi[u] + pj[u] > (NPROD == 6 ? 2 : 4)) continue;
        for (int kb = 0; kb < 2; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][pi[u]], b[kb][pj[u]], acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) C[(lq * 4 + r) * 16 + lr] = acc[r];
}

__global__ void accuracy_kernel(const float* A, const float* B, float* C) {
    const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int k = 0; k < 64; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[lr * 64 + k + lq], B[(k + lq) * 16 + lr], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(lq * 4 + r) * 16 + lr] = acc[r];
    gemm_split<0, 6, 0>(A, B, C + 256);
    gemm_split<0, 6, 1>(A, B, C + 512);
    gemm_split<1, 6, 0>(A, B, C + 768);
    gemm_split<0, 9, 0>(A, B, C + 1024);
    gemm_split<1, 9, 0>(A, B, C + 1280);
}

// lane l supplies the byte address base + group(l>>4) * gstride + (i / 4) * rstride + (i % 4) * 8 with i = l & 15;
// expectation: lane l receives elements (row j = 0..3, column l & 15) of its group's [4][16] block
__global__ void tr_kernel(int* out, int gstride, int rstride) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15;
    const int byte = (l >> 4) * gstride + (i / 4) * rstride + (i % 4) * 8;
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)lds + byte));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)t[j];
}

#define MFMA_ASM(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
template <int CLS>
__device__ __forceinline__ void valu1(float& v, unsigned& n, f32x2& p) {
    if (CLS == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(1.0001f), "v"(0.5f));
    else if (CLS == 1) asm volatile("v_and_b32 %0, %1, %0" : "+v"(n) : "v"(0xffff0000u));
    else if (CLS == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(n) : "v"(n), "v"(0x07060302u));
    else if (CLS == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(n) : "v"(v));
    else if (CLS == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
    else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(f32x2{1.0001f, 1.0001f}), "v"(f32x2{0.5f, 0.5f}));
}
// WAVES waves per SIMD (blockDim = 256 * WAVES), every wave: iters x 8 x { 1 MFMA, K vector instructions of class CLS }
template <int CLS, int K, int WAVES, bool WITH_MFMA>
__global__ void __launch_bounds__(256 * WAVES) issue_kernel(int iters, float* out, long long* cyc) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(1.0f + e * 0.01f); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[6]; unsigned n[6]; f32x2 p[6];
    for (int i = 0; i < 6; ++i) { v[i] = threadIdx.x * 1e-3f + i; n[i] = threadIdx.x + i; p[i] = f32x2{v[i], v[i]}; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (WITH_MFMA) MFMA_ASM(acc[u & 3])
#pragma unroll
            for (int k = 0; k < K; ++k) valu1<CLS>(v[k], n[k], p[k]);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][3];
    for (int i = 0; i < 6; ++i) r += v[i] + n[i] + p[i][0];
    if (r == 12345.678f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CLS, int K, int WAVES, bool WITH_MFMA>
double run_issue(float* out, long long* cyc, int iters) {
    long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((issue_kernel<CLS, K, WAVES, WITH_MFMA>), dim3(256), dim3(256 * WAVES), 0, 0, iters, out, cyc);
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    }
    return (double)h / (iters * 8.0);        // s_memtime ticks (100 MHz on gfx9? printed raw) per MFMA slot of one wave
}
template <int CLS, int WAVES>
void issue_row(const char* name, float* out, long long* cyc) {
    const int it = 4000;
    const double m0 = run_issue<CLS, 0, WAVES, true>(out, cyc, it);
    printf("%-20s %d wave/SIMD  ticks per {MFMA + k ops}: k=0 %.2f | k=1 %.2f  k=2 %.2f  k=3 %.2f  k=4 %.2f  k=6 %.2f | ops alone k=3 %.2f  k=6 %.2f\n",
           name, WAVES, m0, run_issue<CLS, 1, WAVES, true>(out, cyc, it), run_issue<CLS, 2, WAVES, true>(out, cyc, it),
           run_issue<CLS, 3, WAVES, true>(out, cyc, it), run_issue<CLS, 4, WAVES, true>(out, cyc, it),
           run_issue<CLS, 6, WAVES, true>(out, cyc, it), run_issue<CLS, 3, WAVES, false>(out, cyc, it),
           run_issue<CLS, 6, WAVES, false>(out, cyc, it));
}

int main() {
    // ---- A: accuracy -------------------------------------------------------------------------------------------------
    float *dA, *dB, *dC;
    hipMalloc(&dA, 16 * 64 * 4); hipMalloc(&dB, 64 * 16 * 4); hipMalloc(&dC, 6 * 256 * 4);
    const char* names[6] = {"fp32 MFMA 16x16x4", "6 prod trunc, small first", "6 prod trunc, large first", "6 prod RNE, small first",
                            "9 prod trunc", "9 prod RNE"};
    for (int dist = 0; dist < 3; ++dist) {
        double worst[6] = {0}, mean[6] = {0};
        const int trials = 200;
        srand(1234 + dist);
        for (int t = 0; t < trials; ++t) {
            std::vector<float> A(16 * 64), B(64 * 16), C(6 * 256);
            auto rnd = [&]() {
                const double u = rand() / (double)RAND_MAX * 2 - 1;
                if (dist == 0) return (float)u;                                        // uniform [-1, 1)
                if (dist == 1) return (float)(u * std::pow(2.0, (rand() % 24) - 12));   // magnitudes over 2^-12 .. 2^11
                return (float)(std::tanh(2.5 * u));                                     // activation-like
            };
            for (auto& x : A) x = rnd();
            for (auto& x : B) x = rnd();
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(accuracy_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC);
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double ref = 0, mag = 0;
                    for (int k = 0; k < 64; ++k) { ref += (double)A[i * 64 + k] * B[k * 16 + j]; mag += std::fabs((double)A[i * 64 + k] * B[k * 16 + j]); }
                    for (int v = 0; v < 6; ++v) {
                        const double e = std::fabs(C[v * 256 + i * 16 + j] - ref) / mag;
                        worst[v] = std::fmax(worst[v], e);
                        mean[v] += e / (trials * 256.0);
                    }
                }
        }
        printf("A accuracy, distribution %d (|C - fp64| / sum|a b|, K = 64):\n", dist);
        for (int v = 0; v < 6; ++v) printf("   %-28s max %.3e  mean %.3e\n", names[v], worst[v], mean[v]);
    }
    // ---- B: transpose read -------------------------------------------------------------------------------------------
    int* dT; hipMalloc(&dT, 256 * 4);
    const int gs[2] = {128, 1024}, rs[2] = {32, 144};
    for (int c = 0; c < 2; ++c) {
        hipLaunchKernelGGL(tr_kernel, dim3(1), dim3(64), 0, 0, dT, gs[c], rs[c]);
        int h[256]; hipMemcpy(h, dT, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int want = ((l >> 4) * gs[c] + j * rs[c] + (l & 15) * 2) / 2;
                if (h[l * 4 + j] != want) ++bad;
            }
        printf("B ds_read_b64_tr_b16, group stride %d B, row stride %d B: %d of 256 elements differ from (row j, column lane&15); lane 5: %d %d %d %d, lane 21: %d %d %d %d\n",
               gs[c], rs[c], bad, h[20], h[21], h[22], h[23], h[84], h[85], h[86], h[87]);
    }
    // ---- C: issue ----------------------------------------------------------------------------------------------------
    float* out; long long* cyc; hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    printf("C issue (s_memtime ticks; k vector instructions behind every v_mfma_f32_16x16x32_bf16, 4 accumulators):\n");
    issue_row<0, 1>("v_fma_f32", out, cyc);         issue_row<0, 2>("v_fma_f32", out, cyc);
    issue_row<1, 1>("v_and_b32", out, cyc);         issue_row<1, 2>("v_and_b32", out, cyc);
    issue_row<2, 1>("v_perm_b32", out, cyc);        issue_row<2, 2>("v_perm_b32", out, cyc);
    issue_row<3, 1>("v_cvt_pk_bf16_f32", out, cyc); issue_row<3, 2>("v_cvt_pk_bf16_f32", out, cyc);
    issue_row<4, 1>("v_exp_f32", out, cyc);         issue_row<4, 2>("v_exp_f32", out, cyc);
    issue_row<5, 1>("v_pk_fma_f32", out, cyc);      issue_row<5, 2>("v_pk_fma_f32", out, cyc);
    return 0;
}
