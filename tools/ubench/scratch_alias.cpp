// scratch_alias.cpp -- does every wave keep its OWN private (scratch) memory when several workgroups share a CU?
// Context: tile-kernel builds planned with two workgroups per CU returned slightly wrong,
// run-to-run different gradients whenever hipcc had spilled a register. In the smallest such build the ONE spilled value is the
// lane's base pointer of the point prefetch -- stored to scratch once in the prologue, reloaded once per tile -- so a wrong result
// means the reload did not return what the same lane had stored. This probe does the same thing without the rest of the kernel:
// every lane stores a tag (block, thread, slot) into a private array the compiler cannot keep in registers, works for a while
// (LDS + global traffic, barriers, so that co-resident workgroups interleave), reads the tags back and counts mismatches.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/scratch_alias.cpp -o /tmp/scratch_alias && /tmp/scratch_alias
// Output: one line per (threads, dynamic LDS, private floats, workgroups per CU, launch bounds), mismatches and whose tag was read.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Report { unsigned long long bad; unsigned int first_want, first_got, blocks_seen; };

// MODE 0: the private array through ordinary (volatile) accesses -- hipcc addresses it through the flat aperture;
// MODE 1: what a register spill looks like: scratch_store_dword / scratch_load_dword with the architected scratch base and an
// immediate offset (inline asm; the array only reserves the bytes), slots 160 .. 172 as in the tile kernels
template <int OFF> __device__ __forceinline__ void spill_store(unsigned v) {
    asm volatile("scratch_store_dword off, %0, off offset:%1" :: "v"(v), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ unsigned spill_load() {
    unsigned r;
    asm volatile("scratch_load_dword %0, off, off offset:%1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "n"(OFF) : "memory");
    return r;
}
template <int THREADS, int NPRIV, int WPS, int MODE>
__global__ void __launch_bounds__(THREADS, WPS) probe(const float* __restrict__ in, float* __restrict__ out, Report* rep, int rounds,
                                                       const int* __restrict__ perm) {
    extern __shared__ float lds[];
    volatile unsigned int priv[NPRIV];
    const unsigned int tag0 = ((unsigned)blockIdx.x << 16) | ((unsigned)threadIdx.x << 6);
    if (MODE == 0) { for (int i = 0; i < NPRIV; ++i) priv[perm[i]] = tag0 | (unsigned)perm[i]; }
    else { priv[perm[0] & 1] = tag0; spill_store<160>(tag0 | 40u); spill_store<164>(tag0 | 41u); spill_store<168>(tag0 | 42u); }
    float acc = 0.0f;
    for (int r = 0; r < rounds; ++r) {
        // some LDS and global work with barriers in between: lets the workgroups of a CU drift against each other
        for (int i = threadIdx.x; i < 4096; i += THREADS) lds[i] = in[(blockIdx.x * 4096 + i + r) & 0xfffff];
        __syncthreads();
        for (int i = 0; i < 64; ++i) acc = fmaf(lds[(threadIdx.x * 17 + i * 33 + r) & 4095], 1.0001f, acc);
        __syncthreads();
        // the reload: every tag must still be this lane's own
        unsigned int got, want;
        if (MODE == 0) {
            const int slot = perm[r % NPRIV];
            got = priv[slot]; want = tag0 | (unsigned)slot;
            priv[slot] = want;          // (stores keep happening as spills would)
        } else if (r % 3 == 0) { got = spill_load<160>(); want = tag0 | 40u; }
        else if (r % 3 == 1) { got = spill_load<164>(); want = tag0 | 41u; }
        else { got = spill_load<168>(); want = tag0 | 42u; if ((r & 15) == 5) spill_store<168>(want); }
        if (got != want) {
            if (atomicAdd(&rep->bad, 1ull) == 0) { rep->first_want = want; rep->first_got = got; }
        }
    }
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
    if (threadIdx.x == 0) atomicAdd(&rep->blocks_seen, 1u);
}

template <int THREADS, int NPRIV, int WPS, int MODE>
void run(int n_cu, int per_cu, size_t smem, const float* in, float* out, Report* rep, const int* perm, int rounds) {
    CHECK(hipMemset(rep, 0, sizeof(Report)));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<THREADS, NPRIV, WPS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe<THREADS, NPRIV, WPS, MODE>, THREADS, smem));
    const int grid = n_cu * per_cu;
    unsigned long long total_bad = 0;
    Report h{};
    for (int rep_i = 0; rep_i < 5; ++rep_i) {
        hipLaunchKernelGGL((probe<THREADS, NPRIV, WPS, MODE>), dim3(grid), dim3(THREADS), smem, 0, in, out, rep, rounds, perm);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost));
        total_bad = h.bad;
    }
    hipFuncAttributes attr;
    CHECK(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&probe<THREADS, NPRIV, WPS, MODE>)));
    printf("%s threads %4d  lds %6zu B  private %4zu B (kernel: %zu B scratch, %d regs)  launch_bounds(.,%d)  occupancy API %d  grid %5d = %d CUs x %d : "
           "mismatches %llu of %llu reloads", MODE ? "spill-asm " : "flat-array", THREADS, smem, NPRIV * sizeof(float), (size_t)attr.localSizeBytes, attr.numRegs, WPS, occ, grid, n_cu,
           per_cu, total_bad, 5ull * grid * THREADS * rounds);
    if (total_bad) printf("  first: wanted block %u thread %u slot %u, got block %u thread %u slot %u", h.first_want >> 16, (h.first_want >> 6) & 1023,
                          h.first_want & 63, h.first_got >> 16, (h.first_got >> 6) & 1023, h.first_got & 63);
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("%s, %d CUs\n", prop.gcnArchName, n_cu);
    float *in, *out; Report* rep; int* perm;
    CHECK(hipMalloc(&in, (1 << 20) * sizeof(float)));
    CHECK(hipMalloc(&out, (size_t)n_cu * 8 * 1024 * sizeof(float)));
    CHECK(hipMalloc(&rep, sizeof(Report)));
    CHECK(hipMalloc(&perm, 64 * sizeof(int)));
    std::vector<float> hin(1 << 20);
    for (size_t i = 0; i < hin.size(); ++i) hin[i] = (float)(i % 977) * 1e-3f;
    CHECK(hipMemcpy(in, hin.data(), hin.size() * sizeof(float), hipMemcpyHostToDevice));
    int hperm[64];
    for (int i = 0; i < 64; ++i) hperm[i] = i;
    const int rounds = 400;
    for (int npriv_case = 0; npriv_case < 2; ++npriv_case) {
        // permutation within the first NPRIV slots
        const int np = npriv_case == 0 ? 44 : 4;
        for (int i = 0; i < np; ++i) hperm[i] = (i * 7 + 3) % np;
        CHECK(hipMemcpy(perm, hperm, sizeof(hperm), hipMemcpyHostToDevice));
        for (int per_cu = 1; per_cu <= 4; ++per_cu) {
            const size_t smem = 64 * 1024 / (per_cu > 2 ? 2 : 1);
            if (np == 44) {
                run<256, 44, 2, 0>(n_cu, per_cu, smem, in, out, rep, perm, rounds);       // the geometry of the tile kernels: 4 waves, 176 B of scratch
                run<256, 44, 2, 1>(n_cu, per_cu, smem, in, out, rep, perm, rounds);
                run<512, 44, 2, 1>(n_cu, per_cu > 2 ? 2 : per_cu, smem, in, out, rep, perm, rounds);
                run<64, 44, 2, 1>(n_cu, per_cu, smem, in, out, rep, perm, rounds);
            } else {
                run<256, 4, 2, 0>(n_cu, per_cu, smem, in, out, rep, perm, rounds);
                run<128, 4, 2, 0>(n_cu, per_cu, smem, in, out, rep, perm, rounds);
            }
        }
    }
    return 0;
}
