// mfma_lds_hazard.cpp -- is an MFMA operand register safe against the LDS load issued right behind / right in front of it when
// ANOTHER wave of the same SIMD keeps the matrix pipe busy?
// Context (DESIGN.md "two workgroups per CU"): the only kernel of the library that ever returned run-to-run different gradients is the
// split-bf16 kernel of BASELINE config 2 when two INDEPENDENT workgroups share a CU (or its two teams synchronise through LDS flags
// instead of s_barrier), i.e. when the two waves of a SIMD drift freely. Its errors are tiny (loss +-2 ulp, gradients 1e-5): the size of
// the LOW bf16 plane of a split operand. In its weight-gradient loop hipcc emits
//     s_waitcnt lgkmcnt(N)
//     v_mfma_f32_16x16x32_bf16 acc, v[230:233], ...      <- low plane: loaded LAST (RAW right behind the wait), used ONCE ...
//     ds_read_b64_tr_b16 v[230:231], ...                 <- ... and its registers are the destination of the very next LDS load (WAR)
// Neither needs a software wait state according to hipcc's hazard tables. This probe runs both patterns with known operands
// (every k slot 1.0 or 2.0, so that an operand that arrived late or was overwritten early changes the sum) beside a second
// workgroup on the same CU that keeps the matrix pipe busy at a drifting phase.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_lds_hazard.cpp -o /tmp/mfma_lds_hazard && /tmp/mfma_lds_hazard
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Report { unsigned long long wrong, checked; float first_got, first_want; };

// MODE 0: WAR -- MFMA reads v[8:11], the next instruction loads new contents into v[8:11]; the RAW side is padded with 16 wait states
// MODE 1: RAW -- the MFMA follows `s_waitcnt lgkmcnt(0)` on its operand directly; register sets alternate, so the WAR distance is a whole step
// MODE 2: both, as hipcc emits them (no padding anywhere)
// TR: ds_read_b64_tr_b16 (what the kernel uses) or plain ds_read_b64
// NOISE 1: odd workgroups do not measure: they issue MFMA bursts and VALU stretches of changing length (the other wave of the SIMD)
template <int MODE, bool TR, int NOISE>
__global__ void __launch_bounds__(256, 2) probe(Report* rep, int rounds, int pairs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // two regions of constant bf16: X = 1.0 (0x3F80), Y = 2.0 (0x4000); any (transposed) 8-byte read of a region returns four of its value
    unsigned* w = reinterpret_cast<unsigned*>(lds);
    for (int i = threadIdx.x; i < 2048; i += 256) { w[i] = 0x3F803F80u; w[2048 + i] = 0x40004000u; }
    __syncthreads();
    const unsigned ax = (threadIdx.x & 63) * 8, ay = 8192 + (threadIdx.x & 63) * 8;
    const bool noisy = NOISE && (blockIdx.x & 1);
    unsigned long long wrong = 0;
    float fg = 0.f, fw = 0.f;
    unsigned seed = blockIdx.x * 2654435761u + threadIdx.x / 64;
    for (int r = 0; r < rounds; ++r) {
        seed = seed * 1664525u + 1013904223u;
        const int stretch = (seed >> 20) & 63;                  // wave-uniform
        if (noisy) {
            // bursts of 12 independent-accumulator MFMAs (the forward GEMM of the kernel), then a VALU stretch
            for (int b = 0; b < 8 + stretch / 8; ++b)
                asm volatile("v_mfma_f32_16x16x32_bf16 v[32:35], v[20:23], v[24:27], v[32:35]\n\t"
                             "v_mfma_f32_16x16x32_bf16 v[36:39], v[20:23], v[24:27], v[36:39]\n\t"
                             "v_mfma_f32_16x16x32_bf16 v[32:35], v[20:23], v[24:27], v[32:35]\n\t"
                             "v_mfma_f32_16x16x32_bf16 v[36:39], v[20:23], v[24:27], v[36:39]\n\t"
                             "v_mfma_f32_16x16x32_bf16 v[32:35], v[20:23], v[24:27], v[32:35]\n\t"
                             "v_mfma_f32_16x16x32_bf16 v[36:39], v[20:23], v[24:27], v[36:39]\n\t"
                             "v_mfma_f32_16x16x32_bf16 v[32:35], v[20:23], v[24:27], v[32:35]\n\t"
                             "v_mfma_f32_16x16x32_bf16 v[36:39], v[20:23], v[24:27], v[36:39]\n\t"
                             "v_mfma_f32_16x16x32_bf16 v[32:35], v[20:23], v[24:27], v[32:35]\n\t"
                             "v_mfma_f32_16x16x32_bf16 v[36:39], v[20:23], v[24:27], v[36:39]\n\t"
                             "v_mfma_f32_16x16x32_bf16 v[32:35], v[20:23], v[24:27], v[32:35]\n\t"
                             "v_mfma_f32_16x16x32_bf16 v[36:39], v[20:23], v[24:27], v[36:39]\n\t"
                             ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
            float t = (float)r;
            for (int i = 0; i < stretch * 4; ++i) t = fmaf(t, 1.0001f, 0.5f);
            if (t == 12345.0f) w[4096] = 1;
            continue;
        }
        {
            float t = (float)r;
            for (int i = 0; i < stretch; ++i) t = fmaf(t, 1.0001f, 0.5f);          // drift against the neighbours
            if (t == 12345.0f) w[4096] = 1;
        }
        float acc0;
        // (each loop lives in ONE asm statement with fixed registers: the patterns are about exact instruction adjacency)
#define HEAD "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\t" \
             "v_mov_b32 v12, 0x3f803f80\n\tv_mov_b32 v13, 0x3f803f80\n\tv_mov_b32 v14, 0x3f803f80\n\tv_mov_b32 v15, 0x3f803f80\n\t" \
             "s_mov_b32 s20, %3\n\t"
#define TAIL "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\t" \
             "s_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, v0\n\t"
#define MFMA(a) "v_mfma_f32_16x16x32_bf16 v[0:3], " a ", v[12:15], v[0:3]\n\t"
#define LOAD(LD, lo, hi, addr) LD " " lo ", " addr "\n\t" LD " " hi ", " addr "\n\ts_waitcnt lgkmcnt(0)\n\t"
#define PAD "s_nop 7\n\ts_nop 7\n\t"
#define WAR_LOOP(LD) HEAD LOAD(LD, "v[8:9]", "v[10:11]", "%1") PAD "1:\n\t" \
                     MFMA("v[8:11]") LOAD(LD, "v[8:9]", "v[10:11]", "%2") PAD MFMA("v[8:11]") LOAD(LD, "v[8:9]", "v[10:11]", "%1") PAD TAIL
#define RAW_LOOP(LD) HEAD "1:\n\t" LOAD(LD, "v[8:9]", "v[10:11]", "%1") MFMA("v[8:11]") LOAD(LD, "v[16:17]", "v[18:19]", "%2") MFMA("v[16:19]") TAIL
#define BOTH_LOOP(LD) HEAD LOAD(LD, "v[8:9]", "v[10:11]", "%1") "1:\n\t" \
                      MFMA("v[8:11]") LOAD(LD, "v[8:9]", "v[10:11]", "%2") MFMA("v[8:11]") LOAD(LD, "v[8:9]", "v[10:11]", "%1") TAIL
#define RUN(BODY) asm volatile(BODY : "=v"(acc0) : "v"(ax), "v"(ay), "s"(pairs) \
                               : "v0", "v1", "v2", "v3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "s20", "scc", "memory")
        if (MODE == 0) { if (TR) RUN(WAR_LOOP("ds_read_b64_tr_b16")); else RUN(WAR_LOOP("ds_read_b64")); }
        else if (MODE == 1) { if (TR) RUN(RAW_LOOP("ds_read_b64_tr_b16")); else RUN(RAW_LOOP("ds_read_b64")); }
        else { if (TR) RUN(BOTH_LOOP("ds_read_b64_tr_b16")); else RUN(BOTH_LOOP("ds_read_b64")); }
        const float want = 96.0f * (float)pairs;                 // every pair: 32 k slots of 1.0 * 1.0, then 32 of 2.0 * 1.0
        if (acc0 != want) { if (!wrong) { fg = acc0; fw = want; } ++wrong; }
    }
    if (!noisy) {
        if (wrong && atomicAdd(&rep->wrong, wrong) == 0) { rep->first_got = fg; rep->first_want = fw; }
        atomicAdd(&rep->checked, (unsigned long long)rounds);
    }
}

template <int MODE, bool TR, int NOISE>
void run(const char* what, int n_cu, int per_cu, Report* rep) {
    CHECK(hipMemset(rep, 0, sizeof(Report)));
    const size_t smem = 68 * 1024;                                // two workgroups per CU at most, like the kernel
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, TR, NOISE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((probe<MODE, TR, NOISE>), dim3(n_cu * per_cu), dim3(256), smem, 0, rep, 300, 64);
    CHECK(hipDeviceSynchronize());
    Report h;
    CHECK(hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-52s %s  %s  %d WG/CU: %llu of %llu sums wrong", what, TR ? "ds_read_b64_tr_b16" : "ds_read_b64       ",
           NOISE ? "beside MFMA bursts of another workgroup" : "all workgroups alike                   ", per_cu, h.wrong, h.checked);
    if (h.wrong) printf("  (first: %.1f instead of %.1f)", h.first_got, h.first_want);
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("%s, %d CUs; every line: 256-thread workgroups, 300 rounds x 64 (1.0, 2.0) operand pairs per wave\n", prop.gcnArchName, n_cu);
    Report* rep;
    CHECK(hipMalloc(&rep, sizeof(Report)));
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        run<0, true, 0>("WAR: MFMA source = destination of the next LDS load", n_cu, per_cu, rep);
        run<1, true, 0>("RAW: MFMA right behind the wait for its operand", n_cu, per_cu, rep);
        run<2, true, 0>("both, as compiled", n_cu, per_cu, rep);
        run<2, false, 0>("both, as compiled", n_cu, per_cu, rep);
    }
    run<0, true, 1>("WAR: MFMA source = destination of the next LDS load", n_cu, 2, rep);
    run<1, true, 1>("RAW: MFMA right behind the wait for its operand", n_cu, 2, rep);
    run<2, true, 1>("both, as compiled", n_cu, 2, rep);
    run<0, false, 1>("WAR: MFMA source = destination of the next LDS load", n_cu, 2, rep);
    run<1, false, 1>("RAW: MFMA right behind the wait for its operand", n_cu, 2, rep);
    return 0;
}
