// vmcnt_order.cpp -- does s_waitcnt vmcnt(N) mean what hipcc assumes on gfx950?
// hipcc waits for a vector-memory LOAD with `s_waitcnt vmcnt(N)`, N = the vector-memory instructions issued after it (loads AND
// stores: gfx9 has one counter for both) -- correct only if the counter retires in issue order. The tile kernels prefetch (points of
// the next tile, weights of the next layer) far ahead of the use, with dozens of fire-and-forget slab stores in between; builds in
// which such a prefetch "arrived wrong" were cured by -amdgpu-waitcnt-forcezero (pinn_kernel.h). This probe
// asks the hardware directly: one slow load (cold line of a 1 GB buffer), then K fast younger instructions of one kind, then
// `s_waitcnt vmcnt(K)` and a copy of the destination register: if the copy is not the loaded value, younger instructions retired
// the counter before the older load had written its register.
//   kinds of younger instruction: global stores (L2 hits), scratch stores (a spill), global loads from one hot line, scratch loads
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/vmcnt_order.cpp -o /tmp/vmcnt_order && /tmp/vmcnt_order
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Report { unsigned long long early_wrong, checked; unsigned int first_early, first_final; };

// KIND 0: global_store_dword x 8   1: scratch_store_dword x 8   2: global_load_dword (hot line) x 8   3: scratch_load_dword x 8
template <int KIND, int THREADS>
__global__ void __launch_bounds__(THREADS, 2) probe(const unsigned* __restrict__ big, unsigned* sink, const unsigned* hot, Report* rep, int iters,
                                                     unsigned long long mask) {
    extern __shared__ float lds[];
    volatile unsigned int keep[48];          // reserves the private segment the scratch instructions below address
    keep[threadIdx.x & 1] = 1u;
    const unsigned gid = blockIdx.x * THREADS + threadIdx.x;
    unsigned* my_sink = sink + (size_t)gid * 16;
    const unsigned* my_hot = hot + (threadIdx.x & 15);
    unsigned long long wrong = 0;
    unsigned fe = 0, ff = 0;
    asm volatile("scratch_store_dword off, %0, off offset:160\n\ts_waitcnt vmcnt(0)" :: "v"(gid) : "memory");
    for (int it = 0; it < iters; ++it) {
        // a line nobody has touched recently: the load goes all the way to HBM
        const unsigned long long a = ((unsigned long long)gid * 0x9E3779B97F4A7C15ull + (unsigned long long)it * 0xD1B54A32D192ED03ull) & mask;
        const unsigned* p = big + a;
        unsigned v = 0xdeadbeefu, early = 0, t0 = it, t1, t2, t3, t4, t5, t6, t7;
        if (KIND == 0) {
            asm volatile("global_load_dword %0, %2, off\n\t"
                         "global_store_dword %3, %4, off\n\t" "global_store_dword %3, %4, off offset:4\n\t"
                         "global_store_dword %3, %4, off offset:8\n\t" "global_store_dword %3, %4, off offset:12\n\t"
                         "global_store_dword %3, %4, off offset:16\n\t" "global_store_dword %3, %4, off offset:20\n\t"
                         "global_store_dword %3, %4, off offset:24\n\t" "global_store_dword %3, %4, off offset:28\n\t"
                         "s_waitcnt vmcnt(8)\n\t" "v_mov_b32 %1, %0\n\t" "s_waitcnt vmcnt(0)"
                         : "+v"(v), "=&v"(early) : "v"(p), "v"(my_sink), "v"(t0) : "memory");
        } else if (KIND == 1) {
            asm volatile("global_load_dword %0, %2, off\n\t"
                         "scratch_store_dword off, %3, off offset:160\n\t" "scratch_store_dword off, %3, off offset:164\n\t"
                         "scratch_store_dword off, %3, off offset:168\n\t" "scratch_store_dword off, %3, off offset:172\n\t"
                         "scratch_store_dword off, %3, off offset:176\n\t" "scratch_store_dword off, %3, off offset:180\n\t"
                         "scratch_store_dword off, %3, off offset:184\n\t" "scratch_store_dword off, %3, off offset:188\n\t"
                         "s_waitcnt vmcnt(8)\n\t" "v_mov_b32 %1, %0\n\t" "s_waitcnt vmcnt(0)"
                         : "+v"(v), "=&v"(early) : "v"(p), "v"(t0) : "memory");
        } else if (KIND == 2) {
            asm volatile("global_load_dword %0, %10, off\n\t"
                         "global_load_dword %2, %11, off\n\t" "global_load_dword %3, %11, off offset:64\n\t"
                         "global_load_dword %4, %11, off offset:128\n\t" "global_load_dword %5, %11, off offset:192\n\t"
                         "global_load_dword %6, %11, off offset:256\n\t" "global_load_dword %7, %11, off offset:320\n\t"
                         "global_load_dword %8, %11, off offset:384\n\t" "global_load_dword %9, %11, off offset:448\n\t"
                         "s_waitcnt vmcnt(8)\n\t" "v_mov_b32 %1, %0\n\t" "s_waitcnt vmcnt(0)"
                         : "+v"(v), "=&v"(early), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "v"(p), "v"(my_hot) : "memory");
            t0 += t1 + t2 + t3 + t4 + t5 + t6 + t7;
        } else {
            asm volatile("global_load_dword %0, %10, off\n\t"
                         "scratch_load_dword %2, off, off offset:160\n\t" "scratch_load_dword %3, off, off offset:160\n\t"
                         "scratch_load_dword %4, off, off offset:160\n\t" "scratch_load_dword %5, off, off offset:160\n\t"
                         "scratch_load_dword %6, off, off offset:160\n\t" "scratch_load_dword %7, off, off offset:160\n\t"
                         "scratch_load_dword %8, off, off offset:160\n\t" "scratch_load_dword %9, off, off offset:160\n\t"
                         "s_waitcnt vmcnt(8)\n\t" "v_mov_b32 %1, %0\n\t" "s_waitcnt vmcnt(0)"
                         : "+v"(v), "=&v"(early), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "v"(p) : "memory");
            t0 += t1 + t2 + t3 + t4 + t5 + t6 + t7;
        }
        if (early != v) { if (wrong == 0) { fe = early; ff = v; } ++wrong; }
        if (t0 == 0x12345u) lds[threadIdx.x] = 1.0f;           // (keeps the younger loads alive)
    }
    if (wrong) {
        if (atomicAdd(&rep->early_wrong, wrong) == 0) { rep->first_early = fe; rep->first_final = ff; }
    }
    atomicAdd(&rep->checked, (unsigned long long)iters);
}

template <int KIND, int THREADS>
void run(const char* what, int grid, const unsigned* big, unsigned* sink, const unsigned* hot, Report* rep, int iters, unsigned long long mask, int per_cu) {
    CHECK(hipMemset(rep, 0, sizeof(Report)));
    const size_t smem = 60 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<KIND, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((probe<KIND, THREADS>), dim3(grid), dim3(THREADS), smem, 0, big, sink, hot, rep, iters, mask);
    CHECK(hipDeviceSynchronize());
    Report h;
    CHECK(hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-46s %d WG/CU x %3d threads: destination read after `s_waitcnt vmcnt(8)` differs from the loaded value in %llu of %llu loads",
           what, per_cu, THREADS, h.early_wrong, h.checked);
    if (h.early_wrong) printf("  (first: 0x%08x instead of 0x%08x)", h.first_early, h.first_final);
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("%s, %d CUs\n", prop.gcnArchName, n_cu);
    const unsigned long long words = 1ull << 28;            // 1 GB of cold lines
    unsigned *big, *sink, *hot; Report* rep;
    CHECK(hipMalloc(&big, words * 4));
    CHECK(hipMemset(big, 0x5a, words * 4));
    CHECK(hipMalloc(&sink, (size_t)n_cu * 4 * 512 * 16 * 4));
    CHECK(hipMalloc(&hot, 4096));
    CHECK(hipMemset(hot, 0, 4096));
    CHECK(hipMalloc(&rep, sizeof(Report)));
    const int iters = 2000;
    for (int per_cu = 1; per_cu <= 4; per_cu *= 2) {
        run<0, 256>("older HBM load, 8 younger global stores", n_cu * per_cu, big, sink, hot, rep, iters, words - 1, per_cu);
        run<1, 256>("older HBM load, 8 younger scratch stores", n_cu * per_cu, big, sink, hot, rep, iters, words - 1, per_cu);
        run<2, 256>("older HBM load, 8 younger loads of a hot line", n_cu * per_cu, big, sink, hot, rep, iters, words - 1, per_cu);
        run<3, 256>("older HBM load, 8 younger scratch loads", n_cu * per_cu, big, sink, hot, rep, iters, words - 1, per_cu);
    }
    return 0;
}
