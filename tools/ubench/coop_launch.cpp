// launch cost of hipLaunchCooperativeKernel against a plain launch, and the cost of a grid-wide barrier through an atomic counter
// (256 persistent workgroups): would folding pinn_reduce_kernel into the tile kernel's tail pay?
// hipcc --offload-arch=gfx950 -O3 coop_launch.cpp -o coop_launch && ./coop_launch
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void plain(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.0f; }
__global__ void second(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 2.0f; }
__global__ void with_grid_sync(float* p) {
    if (threadIdx.x == 0) p[blockIdx.x] += 1.0f;
    cg::this_grid().sync();
    if (threadIdx.x == 0) p[blockIdx.x] += p[(blockIdx.x + 1) % gridDim.x] * 1e-9f;
}
// hand-made grid barrier: arrive on a counter, spin on its value (all workgroups resident by construction)
__global__ void with_counter(float* p, unsigned* counter, unsigned target) {
    if (threadIdx.x == 0) p[blockIdx.x] += 1.0f;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    if (threadIdx.x == 0) p[blockIdx.x] += p[(blockIdx.x + 1) % gridDim.x] * 1e-9f;
}

int main() {
    const int grid = 256, block = 512, reps = 2000;
    float* p; unsigned* c;
    hipMalloc(&p, grid * sizeof(float)); hipMemset(p, 0, grid * sizeof(float));
    hipMalloc(&c, sizeof(unsigned)); hipMemset(c, 0, sizeof(unsigned));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    auto report = [&](const char* what) { hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("%-44s %.2f us per iteration\n", what, ms * 1e3 / reps); };
    for (int i = 0; i < 100; ++i) plain<<<grid, block>>>(p);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) plain<<<grid, block>>>(p);
    report("one plain launch");
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) { plain<<<grid, block>>>(p); second<<<grid, block>>>(p); }
    report("two dependent plain launches");
    void* args[] = {&p};
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchCooperativeKernel((void*)with_grid_sync, dim3(grid), dim3(block), args, 0, 0);
    report("cooperative launch + grid.sync()");
    unsigned target = 0;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) { target += grid; with_counter<<<grid, block>>>(p, c, target); }
    report("plain launch + atomic-counter grid barrier");
    printf("(%s)\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
