// pk_mfma_coexec.cpp -- are packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) and bf16 MFMAs of ANOTHER wave
// of the same SIMD independent of each other?
// Context (DESIGN.md "two workgroups per CU"): the split-bf16 kernel of BASELINE config 2 -- the one translation unit of the library that
// is compiled WITH the SLP vectoriser's packed fp32 operations AND runs its GEMMs on v_mfma_f32_16x16x32_bf16 -- is the only kernel that
// returns run-to-run different results, and only when the two waves of a SIMD drift freely (two independent workgroups per CU, or
// flag-synchronised teams): one wave's vector phase then runs under the other wave's GEMM phase. Register hazards inside a wave are
// ruled out (tools/ubench/mfma_lds_hazard.cpp, the PINN_SP_DRAIN build). This probe puts one wave of each kind on every SIMD of a CU,
// without any synchronisation between them, and lets each check its own arithmetic:
//   waves 0-3: chains of packed fp32 operations against the same chains in scalar v_fma_f32 / v_mul_f32 / v_add_f32
//   waves 4-7: v_mfma_f32_16x16x32_bf16 on constant operands (every step adds exactly 32 to every accumulator element)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_mfma_coexec.cpp -o /tmp/pk_mfma_coexec && /tmp/pk_mfma_coexec
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Report { unsigned long long pk_wrong, pk_checked, mfma_wrong, mfma_checked; float pk_got[2], pk_want[2], mfma_got, mfma_want; };

// PARTNER: 0 = waves 4-7 idle (exit at once), 1 = bf16 MFMA bursts, 2 = fp32 MFMA bursts (v_mfma_f32_16x16x4_f32)
// PK: 1 = packed instructions in waves 0-3, 0 = the scalar forms only (control: the check itself)
template <int PARTNER, int PK>
__global__ void __launch_bounds__(512, 2) probe(Report* rep, int rounds, float c_in, float d_in) {
    const int wave = threadIdx.x >> 6;
    unsigned long long wrong = 0;
    if (wave < 4) {
        f32x2 got_first = {0.f, 0.f}, want_first = {0.f, 0.f};
        for (int r = 0; r < rounds; ++r) {
            const float seed = 1.0f + (float)((threadIdx.x * 131 + r * 7) & 1023) * 0.0009765625f;
            f32x2 x = {seed, seed + 0.25f}, c = {c_in, c_in * 1.5f}, d = {d_in, -d_in};
            float s0 = x[0], s1 = x[1];
            // 48 dependent steps: fma, mul, add -- packed in x, scalar in (s0, s1)
            for (int i = 0; i < 16; ++i) {
                if (PK) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\t"
                                 "v_pk_mul_f32 %0, %0, %1\n\t"
                                 "v_pk_add_f32 %0, %0, %2" : "+v"(x) : "v"(c), "v"(d));
                } else {
                    float a0 = x[0], a1 = x[1];
                    asm volatile("v_fma_f32 %0, %0, %2, %4\n\tv_fma_f32 %1, %1, %3, %5\n\t"
                                 "v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3\n\t"
                                 "v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5"
                                 : "+v"(a0), "+v"(a1) : "v"(c[0]), "v"(c[1]), "v"(d[0]), "v"(d[1]));
                    x[0] = a0; x[1] = a1;
                }
                asm volatile("v_fma_f32 %0, %0, %2, %4\n\tv_fma_f32 %1, %1, %3, %5\n\t"
                             "v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3\n\t"
                             "v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5"
                             : "+v"(s0), "+v"(s1) : "v"(c[0]), "v"(c[1]), "v"(d[0]), "v"(d[1]));
                // keep the values in range: back towards 1
                x[0] = x[0] * 0.5f + 0.5f; x[1] = x[1] * 0.5f + 0.5f;
                s0 = s0 * 0.5f + 0.5f; s1 = s1 * 0.5f + 0.5f;
            }
            if (x[0] != s0 || x[1] != s1) {
                if (!wrong) { got_first = x; want_first = f32x2{s0, s1}; }
                ++wrong;
            }
        }
        if (wrong && atomicAdd(&rep->pk_wrong, wrong) == 0) {
            rep->pk_got[0] = got_first[0]; rep->pk_got[1] = got_first[1]; rep->pk_want[0] = want_first[0]; rep->pk_want[1] = want_first[1];
        }
        atomicAdd(&rep->pk_checked, (unsigned long long)rounds);
        return;
    }
    if (PARTNER == 0) return;
    float got = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float acc0;
        if (PARTNER == 1) {
            asm volatile("v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\t"
                         "v_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"
                         "v_mov_b32 v8, 0x3f803f80\n\tv_mov_b32 v9, 0x3f803f80\n\tv_mov_b32 v10, 0x3f803f80\n\tv_mov_b32 v11, 0x3f803f80\n\t"
                         "s_mov_b32 s20, 16\n\t"
                         "1:\n\t"
                         "v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[8:11], v[0:3]\n\t"
                         "v_mfma_f32_16x16x32_bf16 v[4:7], v[8:11], v[8:11], v[4:7]\n\t"
                         "v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[8:11], v[0:3]\n\t"
                         "v_mfma_f32_16x16x32_bf16 v[4:7], v[8:11], v[8:11], v[4:7]\n\t"
                         "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\t"
                         "s_nop 7\n\ts_nop 7\n\tv_add_f32 %0, v1, v6\n\t"
                         : "=v"(acc0) :: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "s20", "scc");
        } else {
            asm volatile("v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\t"
                         "v_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"
                         "v_mov_b32 v8, 2.0\n\tv_mov_b32 v9, 4.0\n\t"
                         "s_mov_b32 s20, 16\n\t"
                         "1:\n\t"
                         "v_mfma_f32_16x16x4_f32 v[0:3], v8, v9, v[0:3]\n\t"
                         "v_mfma_f32_16x16x4_f32 v[4:7], v8, v9, v[4:7]\n\t"
                         "v_mfma_f32_16x16x4_f32 v[0:3], v8, v9, v[0:3]\n\t"
                         "v_mfma_f32_16x16x4_f32 v[4:7], v8, v9, v[4:7]\n\t"
                         "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\t"
                         "s_nop 7\n\ts_nop 7\n\tv_add_f32 %0, v1, v6\n\t"
                         : "=v"(acc0) :: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "s20", "scc");
        }
        const float want = 2.0f * 32.0f * 32.0f;                    // two accumulators x 32 MFMAs x 32 per MFMA (K = 32 of 1 * 1, or K = 4 of 2 * 4)
        if (acc0 != want) { if (!wrong) got = acc0; ++wrong; }
    }
    if (wrong && atomicAdd(&rep->mfma_wrong, wrong) == 0) { rep->mfma_got = got; rep->mfma_want = 2048.0f; }
    atomicAdd(&rep->mfma_checked, (unsigned long long)rounds);
}

template <int PARTNER, int PK>
void run(const char* what, int n_cu, int per_cu, Report* rep) {
    CHECK(hipMemset(rep, 0, sizeof(Report)));
    hipLaunchKernelGGL((probe<PARTNER, PK>), dim3(n_cu * per_cu), dim3(512), 0, 0, rep, 2000, 1.0009765625f, 0.37109375f);
    CHECK(hipDeviceSynchronize());
    Report h;
    CHECK(hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-64s %d WG/CU: packed chains wrong %llu of %llu", what, per_cu, h.pk_wrong, h.pk_checked);
    if (h.pk_wrong) printf(" (first: (%.9g, %.9g) instead of (%.9g, %.9g))", h.pk_got[0], h.pk_got[1], h.pk_want[0], h.pk_want[1]);
    printf("; MFMA sums wrong %llu of %llu", h.mfma_wrong, h.mfma_checked);
    if (h.mfma_wrong) printf(" (first: %.1f instead of %.1f)", h.mfma_got, h.mfma_want);
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("%s, %d CUs; 512-thread workgroups: waves 0-3 check 48-step fp32 chains (per lane, 2000 rounds), waves 4-7 their MFMA sums\n", prop.gcnArchName, n_cu);
    Report* rep;
    CHECK(hipMalloc(&rep, sizeof(Report)));
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        run<0, 1>("v_pk_{fma,mul,add}_f32 alone on their SIMD", n_cu, per_cu, rep);
        run<1, 0>("scalar fp32 chains beside v_mfma_f32_16x16x32_bf16 (control)", n_cu, per_cu, rep);
        run<1, 1>("v_pk_{fma,mul,add}_f32 beside v_mfma_f32_16x16x32_bf16", n_cu, per_cu, rep);
        run<2, 1>("v_pk_{fma,mul,add}_f32 beside v_mfma_f32_16x16x4_f32", n_cu, per_cu, rep);
    }
    return 0;
}
