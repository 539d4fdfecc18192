// simd_coexec.cpp -- when does an MFMA result become readable if ANOTHER wave of the same SIMD is busy?
// hipcc guards "MFMA writes a VGPR -> VALU reads it" with a fixed number of wait states (no hardware interlock): passes + 2 or 3,
// counted in the reading wave's own issue slots. That is safe only if nothing stretches the MFMA. This probe puts a victim wave and
// a partner wave on every SIMD of a CU (512-thread workgroups: waves w and w + 4 share SIMD w), no synchronisation between them:
//   victim : 32 MFMAs on two accumulators with constant operands (every MFMA adds exactly 32 to every element), then W wait states,
//            then ONE v_add_f32 of an element of each accumulator -- anything but 2048 means the read saw an unfinished accumulator
//   partner: nothing / scalar fp32 VALU / packed fp32 VALU (v_pk_fma_f32) / transcendentals / fp32 MFMAs / bf16 MFMAs / LDS reads
// for the two MFMA kinds the library uses (v_mfma_f32_16x16x4_f32: exact-fp32 kernels; v_mfma_f32_16x16x32_bf16: split kernels) and
// W = 11 (what hipcc inserts for an 8-pass MFMA), 19, 35.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/simd_coexec.cpp -o /tmp/simd_coexec && /tmp/simd_coexec
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Report { unsigned long long wrong, checked; float got; };

#define INIT "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\t" \
             "v_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"
#define LOOP16(BODY) "s_mov_b32 s20, 16\n\t1:\n\t" BODY "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\t"
#define BF16_OPS "v_mov_b32 v8, 0x3f803f80\n\tv_mov_b32 v9, 0x3f803f80\n\tv_mov_b32 v10, 0x3f803f80\n\tv_mov_b32 v11, 0x3f803f80\n\t"
#define BF16_2 "v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[8:11], v[0:3]\n\tv_mfma_f32_16x16x32_bf16 v[4:7], v[8:11], v[8:11], v[4:7]\n\t"
#define F32_OPS "v_mov_b32 v8, 2.0\n\tv_mov_b32 v9, 4.0\n\t"
#define F32_2 "v_mfma_f32_16x16x4_f32 v[0:3], v8, v9, v[0:3]\n\tv_mfma_f32_16x16x4_f32 v[4:7], v8, v9, v[4:7]\n\t"
// the loop's last iteration is followed by s_sub / s_cmp / s_cbranch (3 issue slots), then the padding, then the read
#define PAD8 "s_nop 7\n\t"
#define PAD16 "s_nop 7\n\ts_nop 7\n\t"
#define PAD32 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
#define READ "v_add_f32 %0, v1, v6\n\t"
#define CLOB "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "s20", "scc"

// VICTIM 0: fp32 MFMA, 1: bf16 MFMA.  WAIT 0 / 1 / 2: 11 / 19 / 35 wait states between the last MFMA and the read
template <int VICTIM, int WAIT>
__device__ __forceinline__ float victim_round() {
    float acc;
    if (VICTIM == 0) {
        if (WAIT == 0) asm volatile(INIT F32_OPS LOOP16(F32_2 F32_2) PAD8 READ : "=v"(acc) :: CLOB);
        else if (WAIT == 1) asm volatile(INIT F32_OPS LOOP16(F32_2 F32_2) PAD16 READ : "=v"(acc) :: CLOB);
        else asm volatile(INIT F32_OPS LOOP16(F32_2 F32_2) PAD32 READ : "=v"(acc) :: CLOB);
    } else {
        if (WAIT == 0) asm volatile(INIT BF16_OPS LOOP16(BF16_2 BF16_2) PAD8 READ : "=v"(acc) :: CLOB);
        else if (WAIT == 1) asm volatile(INIT BF16_OPS LOOP16(BF16_2 BF16_2) PAD16 READ : "=v"(acc) :: CLOB);
        else asm volatile(INIT BF16_OPS LOOP16(BF16_2 BF16_2) PAD32 READ : "=v"(acc) :: CLOB);
    }
    return acc;
}

// PARTNER 0 none, 1 scalar fp32 VALU, 2 packed fp32 VALU, 3 transcendentals, 4 fp32 MFMA, 5 bf16 MFMA, 6 LDS reads
template <int VICTIM, int WAIT, int PARTNER>
__global__ void __launch_bounds__(512, 1) probe(Report* rep, int rounds, volatile int* stop) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = 1.0f;
    __syncthreads();
    if (wave < 4) {
        unsigned long long wrong = 0;
        float got = 0.f;
        for (int r = 0; r < rounds; ++r) {
            const float acc = victim_round<VICTIM, WAIT>();
            if (acc != 2048.0f) { if (!wrong) got = acc; ++wrong; }
        }
        if (wrong && atomicAdd(&rep->wrong, wrong) == 0) rep->got = got;
        atomicAdd(&rep->checked, (unsigned long long)rounds);
        return;
    }
    if (PARTNER == 0) return;
    // the partner works for about as long as the victims do (same number of rounds, each a few hundred issue slots)
    float x = 1.0f + threadIdx.x * 1e-3f, y = 0.5f;
    for (int r = 0; r < rounds; ++r) {
        if (PARTNER == 1) {
            for (int i = 0; i < 16; ++i)
                asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %1, %1, %0, %0\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %1, %1, %0, %0\n\t"
                             "v_mul_f32 %0, 0.5, %0\n\tv_mul_f32 %1, 0.5, %1" : "+v"(x), "+v"(y));
        } else if (PARTNER == 2) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 a = {x, y}, b = {y, x};
            for (int i = 0; i < 16; ++i)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %1, %1, %0, %0\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %1, %1, %0, %0\n\t"
                             "v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %1, %1, %0" : "+v"(a), "+v"(b));
            x = a[0] * 1e-30f + 1.0f; y = b[1] * 1e-30f + 0.5f;
        } else if (PARTNER == 3) {
            for (int i = 0; i < 16; ++i)
                asm volatile("v_exp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_exp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_mul_f32 %0, 0.01, %0\n\tv_add_f32 %1, 1.0, %1" : "+v"(x), "+v"(y));
        } else if (PARTNER == 4) {
            asm volatile("v_mov_b32 v20, 2.0\n\tv_mov_b32 v21, 4.0\n\t"
                         LOOP16("v_mfma_f32_16x16x4_f32 v[24:27], v20, v21, v[24:27]\n\tv_mfma_f32_16x16x4_f32 v[28:31], v20, v21, v[28:31]\n\t"
                                "v_mfma_f32_16x16x4_f32 v[24:27], v20, v21, v[24:27]\n\tv_mfma_f32_16x16x4_f32 v[28:31], v20, v21, v[28:31]\n\t")
                         ::: "v20", "v21", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "s20", "scc");
        } else if (PARTNER == 5) {
            asm volatile("v_mov_b32 v20, 0x3f803f80\n\tv_mov_b32 v21, 0x3f803f80\n\tv_mov_b32 v22, 0x3f803f80\n\tv_mov_b32 v23, 0x3f803f80\n\t"
                         LOOP16("v_mfma_f32_16x16x32_bf16 v[24:27], v[20:23], v[20:23], v[24:27]\n\tv_mfma_f32_16x16x32_bf16 v[28:31], v[20:23], v[20:23], v[28:31]\n\t"
                                "v_mfma_f32_16x16x32_bf16 v[24:27], v[20:23], v[20:23], v[24:27]\n\tv_mfma_f32_16x16x32_bf16 v[28:31], v[20:23], v[20:23], v[28:31]\n\t")
                         ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "s20", "scc");
        } else {
            for (int i = 0; i < 32; ++i) x += lds[(threadIdx.x * 4 + i * 64 + (int)y) & 4095];
            y = x * 1e-30f;
        }
    }
    if (x == 123.456f && y == 1.0f) *stop = 1;
}

static Report* g_rep;
static int* g_stop;
static int g_cus;

template <int VICTIM, int WAIT, int PARTNER>
unsigned long long run() {
    CHECK(hipMemset(g_rep, 0, sizeof(Report)));
    hipLaunchKernelGGL((probe<VICTIM, WAIT, PARTNER>), dim3(g_cus), dim3(512), 0, 0, g_rep, 3000, g_stop);
    CHECK(hipDeviceSynchronize());
    Report h;
    CHECK(hipMemcpy(&h, g_rep, sizeof(h), hipMemcpyDeviceToHost));
    printf(" %9llu%s", h.wrong, h.wrong ? "" : " ");
    if (h.wrong) printf("(%.0f)", h.got);
    return h.checked;
}

template <int VICTIM, int WAIT>
void row(const char* what) {
    printf("%-44s", what);
    unsigned long long n = 0;
    n = run<VICTIM, WAIT, 0>(); run<VICTIM, WAIT, 1>(); run<VICTIM, WAIT, 2>(); run<VICTIM, WAIT, 3>();
    run<VICTIM, WAIT, 4>(); run<VICTIM, WAIT, 5>(); run<VICTIM, WAIT, 6>();
    printf("   of %llu\n", n);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    printf("%s, %d CUs; wrong sums (first wrong value) per partner activity on the same SIMD\n", prop.gcnArchName, g_cus);
    CHECK(hipMalloc(&g_rep, sizeof(Report)));
    CHECK(hipMalloc(&g_stop, sizeof(int)));
    printf("%-44s %10s %10s %10s %10s %10s %10s %10s\n", "victim: 32 MFMAs, W wait states, one read", "alone", "fp32 VALU", "v_pk_*_f32",
           "exp / rcp", "fp32 MFMA", "bf16 MFMA", "LDS reads");
    row<0, 0>("v_mfma_f32_16x16x4_f32, W = 11");
    row<0, 1>("v_mfma_f32_16x16x4_f32, W = 19");
    row<0, 2>("v_mfma_f32_16x16x4_f32, W = 35");
    row<1, 0>("v_mfma_f32_16x16x32_bf16, W = 11");
    row<1, 1>("v_mfma_f32_16x16x32_bf16, W = 19");
    row<1, 2>("v_mfma_f32_16x16x32_bf16, W = 35");
    return 0;
}
