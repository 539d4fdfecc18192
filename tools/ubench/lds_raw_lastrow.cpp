// lds_raw_lastrow.cpp -- is the data of an LDS load in the registers of ALL lanes when `s_waitcnt lgkmcnt(N)` lets the wave go on?
// Context (DESIGN.md "two workgroups per CU"): tools/diff_runs.py --net on the builds that return run-to-run different gradients shows,
// in every affected (tile, wave), wrong first-layer values in lanes 48-63 ONLY (the last of the four 16-lane passes of a wave64
// instruction) and ONLY in the low half of a packed pair -- the operand hipcc copies out of a ds_read_b128 destination with a v_mov_b32
// issued directly behind the s_waitcnt that covers that load:
//     ds_read_b128 v[84:87], ...  /  s_waitcnt lgkmcnt(3)  /  v_mov_b32 v62, v84  /  ...  /  v_pk_fma_f32 v[62:63], v[62:63], ...
// and only while the other wave of the SIMD is in a GEMM phase (bf16 MFMAs writing their results back). Inputs dumped later from the same
// registers are right, so the copy ran before the last pass of the load had been written. This probe: alternating loads of two constant
// LDS regions into the same registers, a wait, N wait states, one consumer instruction, the copy checked lane by lane; beside a partner
// wave that issues bf16 MFMA bursts.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_raw_lastrow.cpp -o /tmp/lds_raw_lastrow && /tmp/lds_raw_lastrow
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Report { unsigned long long wrong, checked, by_row[4]; };

// one round: 32 x { load X -> copy -> check-accumulate ; load Y -> copy -> check-accumulate }. The copies are summed (X = 1.0, Y = 2.0 in
// every dword): 32 * 3 = 96 per lane if every copy saw its own load.
#define HEAD "v_mov_b32 v60, 0\n\ts_mov_b32 s20, 32\n\t1:\n\t"
#define TAIL "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\t" "s_nop 7\n\tv_mov_b32 %0, v60\n\t"
// WIDTH 128 / 64 / 32;  consumer: v_mov_b32 (then added), v_add_f32 directly, v_pk_add_f32 (low half = the loaded dword)
#define STEP128(addr, N, CONS) "ds_read_b128 v[84:87], " addr "\n\ts_waitcnt lgkmcnt(0)\n\t" N CONS
#define STEP64(addr, N, CONS) "ds_read_b64 v[84:85], " addr "\n\ts_waitcnt lgkmcnt(0)\n\t" N CONS
#define STEP32(addr, N, CONS) "ds_read_b32 v84, " addr "\n\ts_waitcnt lgkmcnt(0)\n\t" N CONS
#define CONS_MOV "v_mov_b32 v62, v84\n\ts_nop 3\n\tv_add_f32 v60, v60, v62\n\t"
#define CONS_ADD "v_add_f32 v60, v60, v84\n\t"
#define CONS_LAST "v_mov_b32 v62, v87\n\ts_nop 3\n\tv_add_f32 v60, v60, v62\n\t"      /* the LAST dword of the b128 */
#define CLOB "v60", "v62", "v63", "v84", "v85", "v86", "v87", "s20", "scc", "memory"

template <int KIND>
__device__ __forceinline__ float victim_round(unsigned ax, unsigned ay) {
    float acc;
    if (KIND == 0) asm volatile(HEAD STEP128("%1", "", CONS_MOV) STEP128("%2", "", CONS_MOV) TAIL : "=v"(acc) : "v"(ax), "v"(ay) : CLOB);
    else if (KIND == 1) asm volatile(HEAD STEP128("%1", "s_nop 0\n\t", CONS_MOV) STEP128("%2", "s_nop 0\n\t", CONS_MOV) TAIL : "=v"(acc) : "v"(ax), "v"(ay) : CLOB);
    else if (KIND == 2) asm volatile(HEAD STEP128("%1", "s_nop 1\n\t", CONS_MOV) STEP128("%2", "s_nop 1\n\t", CONS_MOV) TAIL : "=v"(acc) : "v"(ax), "v"(ay) : CLOB);
    else if (KIND == 3) asm volatile(HEAD STEP128("%1", "s_nop 3\n\t", CONS_MOV) STEP128("%2", "s_nop 3\n\t", CONS_MOV) TAIL : "=v"(acc) : "v"(ax), "v"(ay) : CLOB);
    else if (KIND == 4) asm volatile(HEAD STEP128("%1", "", CONS_ADD) STEP128("%2", "", CONS_ADD) TAIL : "=v"(acc) : "v"(ax), "v"(ay) : CLOB);
    else if (KIND == 5) asm volatile(HEAD STEP128("%1", "", CONS_LAST) STEP128("%2", "", CONS_LAST) TAIL : "=v"(acc) : "v"(ax), "v"(ay) : CLOB);
    else if (KIND == 6) asm volatile(HEAD STEP64("%1", "", CONS_MOV) STEP64("%2", "", CONS_MOV) TAIL : "=v"(acc) : "v"(ax), "v"(ay) : CLOB);
    else asm volatile(HEAD STEP32("%1", "", CONS_MOV) STEP32("%2", "", CONS_MOV) TAIL : "=v"(acc) : "v"(ax), "v"(ay) : CLOB);
    return acc;
}

// PARTNER 0 none, 1 bf16 MFMA bursts, 2 fp32 VALU
template <int KIND, int PARTNER>
__global__ void __launch_bounds__(512, 1) probe(Report* rep, int rounds, volatile int* stop) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 1024];
    for (int i = threadIdx.x; i < 1024; i += 512) { lds[i] = 1.0f; lds[1024 + i] = 2.0f; }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#ifdef FULL_VGPRS
    asm volatile("v_mov_b32 v255, 0" ::: "v255");      // the kernel then allocates all 256 registers of a wave: two waves fill a SIMD's file
#endif
    if (wave < 4) {
        // lane-private 16-byte slots (conflict-free b128 reads), X region then Y region
#ifdef CONFLICT_READS
        // the kernel's first-layer weight reads: the 16 lanes of a row read ONE 16-byte row (broadcast), the four rows of the wave
        // read addresses 128 bytes apart (same banks: the LDS serves them one after the other, lanes 48-63 last)
        const unsigned ax = (wave * 512 + (lane >> 4) * 128) % 4096, ay = ax + 4096;
#else
        const unsigned ax = ((wave * 64 + lane) * 16) % 4096, ay = ax + 4096;      // (`lds` is the only LDS object: offset 0)
#endif
        unsigned long long wrong = 0;
        for (int r = 0; r < rounds; ++r) {
            const float acc = victim_round<KIND>(ax, ay);
            if (acc != 96.0f) ++wrong;
        }
        if (wrong) { atomicAdd(&rep->wrong, wrong); atomicAdd(&rep->by_row[lane >> 4], wrong); }
        atomicAdd(&rep->checked, (unsigned long long)rounds);
        return;
    }
    if (PARTNER == 0) return;
    float x = 1.0f + threadIdx.x * 1e-3f, y = 0.5f;
    for (int r = 0; r < rounds * 4; ++r) {
        if (PARTNER == 1) {
            asm volatile("v_mov_b32 v20, 0x3f803f80\n\tv_mov_b32 v21, 0x3f803f80\n\tv_mov_b32 v22, 0x3f803f80\n\tv_mov_b32 v23, 0x3f803f80\n\t"
                         "s_mov_b32 s20, 3\n\t1:\n\t"
                         "v_mfma_f32_16x16x32_bf16 v[24:27], v[20:23], v[20:23], v[24:27]\n\tv_mfma_f32_16x16x32_bf16 v[28:31], v[20:23], v[20:23], v[28:31]\n\t"
                         "v_mfma_f32_16x16x32_bf16 v[24:27], v[20:23], v[20:23], v[24:27]\n\tv_mfma_f32_16x16x32_bf16 v[28:31], v[20:23], v[20:23], v[28:31]\n\t"
                         "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\t"
                         ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "s20", "scc");
        } else {
            for (int i = 0; i < 8; ++i)
                asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %1, %1, %0, %0\n\tv_mul_f32 %0, 0.5, %0\n\tv_mul_f32 %1, 0.5, %1" : "+v"(x), "+v"(y));
        }
        float t = x;
        for (int i = 0; i < ((r * 7 + wave) & 7); ++i) t = fmaf(t, 1.0001f, 0.5f);
        if (t == 12345.0f) *stop = 1;
    }
    if (x == 123.456f && y == 1.0f) *stop = 1;
}

static Report* g_rep;
static int* g_stop;
static int g_cus;

template <int KIND, int PARTNER>
void cell() {
    CHECK(hipMemset(g_rep, 0, sizeof(Report)));
    hipLaunchKernelGGL((probe<KIND, PARTNER>), dim3(g_cus), dim3(512), 0, 0, g_rep, 4000, g_stop);
    CHECK(hipDeviceSynchronize());
    Report h;
    CHECK(hipMemcpy(&h, g_rep, sizeof(h), hipMemcpyDeviceToHost));
    printf("   %9llu", h.wrong);
    if (h.wrong) printf(" [lanes 0-15: %llu, 16-31: %llu, 32-47: %llu, 48-63: %llu]", h.by_row[0], h.by_row[1], h.by_row[2], h.by_row[3]);
}

template <int KIND>
void row(const char* what) {
    printf("%-62s", what);
    cell<KIND, 0>(); cell<KIND, 2>(); cell<KIND, 1>();
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    printf("%s, %d CUs; lanes whose 64 copies (per round; 4000 rounds, 262144 lanes) did not all see their own load: alone / beside fp32 VALU / beside bf16 MFMA bursts\n",
           prop.gcnArchName, g_cus);
    CHECK(hipMalloc(&g_rep, sizeof(Report)));
    CHECK(hipMalloc(&g_stop, sizeof(int)));
    row<0>("ds_read_b128; s_waitcnt lgkmcnt(0); v_mov_b32 of dword 0");
    row<1>("ds_read_b128; s_waitcnt lgkmcnt(0); s_nop 0; v_mov_b32");
    row<2>("ds_read_b128; s_waitcnt lgkmcnt(0); s_nop 1; v_mov_b32");
    row<3>("ds_read_b128; s_waitcnt lgkmcnt(0); s_nop 3; v_mov_b32");
    row<4>("ds_read_b128; s_waitcnt lgkmcnt(0); v_add_f32 of dword 0");
    row<5>("ds_read_b128; s_waitcnt lgkmcnt(0); v_mov_b32 of dword 3");
    row<6>("ds_read_b64;  s_waitcnt lgkmcnt(0); v_mov_b32 of dword 0");
    row<7>("ds_read_b32;  s_waitcnt lgkmcnt(0); v_mov_b32");
    return 0;
}
