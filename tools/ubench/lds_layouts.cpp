// LDS layouts for the split-bf16 activation planes (round 3): cycles per wave-instruction of the three accesses the tile kernel
// makes -- ds_write_b64 of a lane's 4 units, ds_read_b128 of 8 units along K, ds_read_b64_tr_b16 of 4 points x 16 units -- under
// candidate address maps, 8 waves per CU all issuing the same access (the bank model of tools/layout/lds_banks.py does not
// cover the transpose read: MI355X_MICROARCH.md "further conflict classes").
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_layouts.cpp -o /tmp/lds_layouts && /tmp/lds_layouts
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one plane: rows r = 0..63 ((stream, point) pairs), 64 units of 2 bytes. -> byte address of the 8-byte slot q (4 units) of row r
template <int L>
__device__ __forceinline__ int slot_addr(int r, int q /*0..15*/) {
    if (L == 0) return r * 128 + ((((q >> 1) ^ (r & 7)) << 4) | ((q & 1) << 3));                 // rows of 128 B, chunk ^ (row & 7)
    if (L == 1) return (q >> 2) * 2048 + r * 32 + (q & 3) * 8;                                   // blocked [16-unit block][row][16 units]
    if (L == 2) return (q >> 2) * 2048 + r * 32 + (((q & 3) ^ (2 * ((r >> 2) & 1))) * 8);        // blocked, slot pair swizzled by row bit 2
    if (L == 3) return r * 160 + q * 8;                                                          // padded rows
    if (L == 4) return r * 144 + q * 8;
    if (L == 5) return (q >> 2) * 2064 + r * 32 + (q & 3) * 8;                                   // blocked, 16 B between blocks
    return (q >> 2) * 2048 + r * 32 + (((q & 3) ^ ((r >> 2) & 3)) * 8);                          // blocked, slot ^ row bits 2..3 (reads permuted)
}

// MODE 0: ds_write_b64 (lane (lr, lq) of wave w: row lr (+16 per instruction), slot 4 w + lq)
// MODE 1: ds_read_b128 (row lr, chunk 4 kb + lq = slots 2 * chunk, 2 * chunk + 1 -- valid where the two are adjacent)
// MODE 2: ds_read_b64_tr_b16, rows 4 lq + lr / 4 of one 16-row stream block, unit block o: slot 4 o + (lr & 3)
// MODE 3: the same with the rows of the T = 16 map of the first kernel version: (lq >> 1) * 16 + 4 (lq & 1) + lr / 4
template <int L, int MODE>
__global__ void __launch_bounds__(512) k(int iters, long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 40960 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, lr = lane & 15, lq = lane >> 4;
    int a[4];
    for (int i = 0; i < 4; ++i) {
        if (MODE == 0) a[i] = slot_addr<L>(lr + 16 * i, 4 * wave + lq);
        else if (MODE == 1) a[i] = slot_addr<L>(lr + 16 * (i & 1), 2 * (4 * (i >> 1) + lq));
        else if (MODE == 2) a[i] = slot_addr<L>(16 * (i & 1) + 4 * lq + (lr >> 2), 4 * ((i >> 1) + wave & 3) + (lr & 3));
        else a[i] = slot_addr<L>(32 * (i & 1) + (lq >> 1) * 16 + 4 * (lq & 1) + (lr >> 2), 4 * ((i >> 1) + wave & 3) + (lr & 3));
    }
    u32x4 acc = {0, 0, 0, 0};
    u32x2 v = {(unsigned)lane, (unsigned)wave};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 0) {
                    asm volatile("ds_write_b64 %0, %1" :: "v"(a[i]), "v"(v) : "memory");
                } else if (MODE == 1) {
                    u32x4 x;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(a[i]) : "memory");
                    asm volatile("" :: "v"(x));
                } else {
                    u32x2 x;
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(x) : "v"(a[i]) : "memory");
                    asm volatile("" :: "v"(x));
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (acc[0] == 12345u) sink[0] = 1.0f;
}

template <int L, int MODE>
double run(long long* cyc, float* sink) {
    const int iters = 2000;
    long long h = 0;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<L, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<L, MODE>), dim3(256), dim3(512), 40960, 0, iters, cyc, sink);
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    }
    return (double)h / (iters * 16.0) / 8.0 * 8.0;       // cycles per wave-instruction of ONE wave (8 waves share the LDS)
}

template <int L>
void row(const char* name, long long* cyc, float* sink) {
    printf("%-44s write_b64 %6.1f   read_b128 %6.1f   tr rows 4lq.. %6.1f   tr first-version rows %6.1f   (cycles per wave-instruction, 8 waves per CU)\n",
           name, run<L, 0>(cyc, sink), run<L, 1>(cyc, sink), run<L, 2>(cyc, sink), run<L, 3>(cyc, sink));
}

int main() {
    long long* cyc; float* sink;
    hipMalloc(&cyc, 8); hipMalloc(&sink, 4);
    row<0>("0 rows 128 B, chunk ^ (row & 7)", cyc, sink);
    row<1>("1 blocked [block][row][16], rows 32 B", cyc, sink);
    row<2>("2 blocked, slot ^ 2 (row >> 2 & 1)", cyc, sink);
    row<3>("3 rows 160 B", cyc, sink);
    row<4>("4 rows 144 B", cyc, sink);
    row<5>("5 blocked, 2064 B per block", cyc, sink);
    row<6>("6 blocked, slot ^ (row >> 2 & 3)", cyc, sink);
    return 0;
}
