// pinn_fit_kernel.h -- a whole chunk of `Solver.fit` iterations in ONE launch (round 5).
//
// The reference's own regime -- batches of 100 .. 1 500 points (tutorials/1. Solving PDEs.ipynb, README.md:50-53), nets of 10 .. 40
// units -- is launch bound on a GPU: two launches per iteration (tile kernel; reduction + Adam + next batch) even when replayed as
// a launch graph, 16 us per iteration for 7 tiles of work. Here the iteration loop of model_torch.py:426-464 runs INSIDE the kernel:
//
//     for k in 0 .. K-1:   draw the points of this workgroup's tiles (Philox, the counters of pinn_sample_kernel)
//                          -> tile body (forward jets .. reverse sweep), partial gradient row of this workgroup
//                          -> ONE device-scope arrive / wait over the (resident) workgroups of the grid
//                          -> every workgroup sums ALL partial rows in the order of pinn_reduce_kernel and applies Adam to ITS OWN
//                             copy of (parameters, exp_avg, exp_avg_sq): no second wait, no broadcast
//
// Every workgroup computes the same sums in the same order, so the copies stay bit-identical; workgroup 0 writes the loss history,
// the gradient buffer of the last iteration and, at the end, the parameters and the Adam state back to the caller's buffers. The
// partial rows are double buffered by iteration parity (a fast workgroup is at most one wait ahead of a slow one). The trajectory is
// the eager loop's (pinn_fit_steps) to fp32 round-off: same tile -> workgroup map (same grid), same summation order, same Adam scalars
// (host doubles through PinnFitCtrl), same Philox counters -- and the same source for the tile pass, but compiled into another kernel:
// hipcc contracts a * b + c into fused multiply-adds where it sees fit, not identically in both, so the last bit may differ (measured:
// 5e-7 relative on the losses after 400 iterations; the launch-graph form of round 4 replays the very same kernels and IS bit-identical).
//
// Widths <= 32 only (one or two waves per workgroup; what fits this regime), grids of at most PINN_FIT_MAX_WGS workgroups, every
// one resident (a grid of <= 64 workgroups of <= 128 threads on a 256-CU device). The wait is bounded: a workgroup that does not see
// its peers within PINN_FIT_SPIN_LIMIT polls raises the error flag and every workgroup leaves (the host then reports it).
#pragma once
#include "pinn_kernel.h"
#include "pinn_aux_kernels.h"

#define PINN_FIT_MAX_WGS 64
#define PINN_FIT_SPIN_LIMIT (1 << 22)

struct PinnFitP {
    const PinnFitCtrl* ctrl;            // step sizes, loss slots, Philox call index and key of the chunk (pinn_fit_ctrl_kernel, launched in front)
    int k_steps;
    float* params; float* m; float* v;  // the caller's buffers: read at the start, written back by workgroup 0 at the end
    const unsigned char* mask;
    int* step_ptr;
    float* grads;
    float* rows;                        // [2][grid][p_core] partial gradient rows, double buffered by iteration parity
    float* state;                       // [grid][3][p_core] (parameters | exp_avg | exp_avg_sq) of every workgroup
    unsigned* sync;                     // [0] arrival counter, [1] error flag; zeroed by the host in front of the launch
    float* xs; long long n; PinnSampleSpec spec;
    float b1, b2, eps;
    int off_loss;
};

#ifndef PINN_EMU
PINN_DEVICE unsigned pinn_fit_arrive_and_wait(unsigned* sync, unsigned target) {
    // release our partial row, acquire everybody else's (agent scope: the rows cross CUs through L2)
    __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned polls = 0;
    while (__hip_atomic_load(sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return 1u;
        if (++polls > PINN_FIT_SPIN_LIMIT) { __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return 1u; }
        __builtin_amdgcn_s_sleep(2);
    }
    return 0u;
}

template <int HP, int ND, int N2, int MT, int LHC, int ACTC, bool COMB, int VAR>
PINN_GLOBAL void PINN_LAUNCH_BOUNDS2((PinnCfg<HP, ND, N2, MT>::NTHREADS * ((VAR & 256) ? 2 : 1)),
                                    (PinnCfg<HP, ND, N2, MT>::NW < 4 || (VAR & (2 | 256)) ? 2 : PINN_WAVES_PER_SIMD))
pinn_fit_kernel(const PinnKArgs A0, const PinnFitP P) {
    using C = PinnCfg<HP, ND, N2, MT, (VAR & 512) != 0>;
    constexpr int TEAMS = (VAR & 256) ? 2 : 1, NTH = C::NTHREADS * TEAMS, T = C::T;
    static_assert(HP <= 32 && !(VAR & (128 | 512 | 64)), "the one-launch fit chunk is built for the narrow nets (no streamed weight gradients)");
    const int tid = PINN_TID, bid = PINN_BID, G = PINN_NBLK;
    const int pc = A0.p_core;
    float* my = P.state + (size_t)bid * 3 * pc;           // parameters | exp_avg | exp_avg_sq of this workgroup
    for (int i = tid; i < pc; i += NTH) { my[i] = P.params[i]; my[pc + i] = P.m[i]; my[2 * pc + i] = P.v[i]; }
    const PinnKArgs& A = A0;
    __shared__ unsigned bail;
    if (tid == 0) bail = 0u;
    __syncthreads();
    const PinnFitCtrl* ctrl = P.ctrl;
    const unsigned k0 = ctrl->k0, k1 = ctrl->k1;
    for (int k = 0; k < P.k_steps; ++k) {
        // (a) the batch of iteration k, the points of this workgroup's tiles only (the tile body reads nothing else)
        const unsigned long long call = ctrl->call_index0 + (unsigned long long)k;
        for (long long tile = A.tile_begin + (long long)bid * TEAMS; tile < A.tile_end; tile += (long long)G * TEAMS) {
            for (long long i = tile * T + tid; i < (tile + TEAMS) * T && i < P.n; i += NTH)
                pinn_sample_point(P.xs, i, P.spec, k0, k1, (unsigned)(call & 0xffffffffull), (unsigned)(call >> 32));
        }
        __threadfence_block();
        __syncthreads();
        // (b) forward jets .. reverse sweep of this workgroup's tiles; its partial row of this iteration's parity
        pinn_tile_body<HP, ND, N2, MT, LHC, ACTC, COMB, VAR>(A, my, P.rows + (size_t)(k & 1) * G * pc);
        // (c) one arrive / wait of the grid
        __syncthreads();
        if (tid == 0) bail = pinn_fit_arrive_and_wait(P.sync, (unsigned)(k + 1) * (unsigned)G);
        __syncthreads();
        if (bail) return;
        // (d) sum of the partial rows in pinn_reduce_kernel's order (chunks of workgroups c = w mod CH, then the chunks in ascending
        //     order: for G <= CH that is row after row), Adam on this workgroup's own copy
        const float* rows = P.rows + (size_t)(k & 1) * G * pc;
        const float step_size = ctrl->step_size[k], bc2_sqrt = ctrl->bc2_sqrt[k];
        // (the G row loads of a parameter are independent: issued eight at a time, summed in ascending row order -- the first form of
        //  this loop waited for one L2 round trip per row and made the iteration 2.6x SLOWER than two launches: 43 us against 16)
        for (int p0 = tid; p0 < pc; p0 += 4 * NTH) {
            float t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int c0 = 0; c0 < G; c0 += 8) {
                float r[8][4];
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int p = p0 + q * NTH, c = c0 + j;
                        r[j][q] = (c < G && p < pc) ? rows[(size_t)c * pc + p] : 0.0f;
                    }
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) t[q] += r[j][q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = p0 + q * NTH;
                if (p >= pc) continue;
                if (bid == 0) {
                    if (k + 1 == P.k_steps) P.grads[p] = t[q];
                    if (p == P.off_loss) ctrl->loss_base[k] = t[q];
                }
                if (!P.mask || P.mask[p]) pinn_adam_update(my, t[q], my + pc, my + 2 * pc, p, step_size, bc2_sqrt, P.b1, P.b2, P.eps);
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (bid == 0) {
        for (int i = tid; i < pc; i += NTH) { P.params[i] = my[i]; P.m[i] = my[pc + i]; P.v[i] = my[2 * pc + i]; }
        if (tid == 0) P.step_ptr[0] = ctrl->step0 + P.k_steps - 1;
    }
}
#endif
