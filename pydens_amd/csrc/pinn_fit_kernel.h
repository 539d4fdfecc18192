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

// PinnFitVw<...>::VW: how many virtual workgroups (pinn_kernel.h, "VW > 1") of an instantiation one CU holds -- at most 8 waves (two per
// SIMD: the 256 registers the narrow kernels are compiled for), and within the 160 KB of LDS: VW blocks, ONE copy of W^T, and what the
// one-CU form keeps resident -- parameters, exp_avg, exp_avg_sq, one partial row per virtual workgroup, the batch and its pre-pass rows
// (priced here for a net of max(LHC, 2) hidden->hidden layers; the launcher checks the real figure and declines if it does not fit)
template <int HP, int ND, int N2, int MT, int LHC, int VAR>
struct PinnFitVw {
    using C = PinnCfg<HP, ND, N2, MT, false>;
    static constexpr int BLOCK = (((VAR >> 4) & 3) != 0) ? C::TEAM_FLOATS : C::SMEM_FLOATS;
    static constexpr int WT = C::wt_fits(LHC) ? LHC * HP * C::WT_LD : 0;
    static constexpr int PC_EST = HP * 8 + HP + (LHC > 2 ? LHC : 2) * (HP * HP + HP) + HP + 16;
    static constexpr int BY_WAVES = 8 / C::NW;
    static constexpr int BY_LDS = (160 * 1024 / 4 - WT - 3 * PC_EST - 1024) / (BLOCK + PC_EST);
    static constexpr int VW = (BY_LDS < BY_WAVES) ? BY_LDS : BY_WAVES;
    static constexpr int BASE_FLOATS = VW * BLOCK + WT;        // the resident arrays start here
    // floats behind BASE_FLOATS: state [3][pcp] | rows [VW][pcp] | batch [n * d] | pre-pass rows [n_aux * n]   (pcp = p_core rounded up to 4)
    PINN_HOST_DEVICE static constexpr long long resident_floats(int p_core, long long n, int d, int n_aux) {
        return (long long)(3 + VW) * ((p_core + 3) / 4 * 4) + (n * d + 3) / 4 * 4 + (n * n_aux + 3) / 4 * 4;
    }
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
#endif

// VW = 1: a grid of G resident workgroups, one device-scope wait per iteration (the first form: 22 - 37 us per iteration on MI355X, the
// wait crosses the XCDs' L2s -- slower than replaying two launches, opt-in). VW > 1: ONE hardware workgroup of VW virtual ones on one CU
// (G = 1): the iteration's only synchronisation is the workgroup barrier -- the form for batches of a few tiles.
template <int HP, int ND, int N2, int MT, int LHC, int ACTC, bool COMB, int VAR, int VW>
PINN_GLOBAL void PINN_LAUNCH_BOUNDS2((PinnCfg<HP, ND, N2, MT>::NTHREADS * VW), 2)
pinn_fit_kernel(const PinnKArgs A0, const PinnFitP P) {
    using C = PinnCfg<HP, ND, N2, MT, (VAR & 512) != 0>;
    constexpr int NTH1 = C::NTHREADS, NTH = NTH1 * VW, T = C::T;
    static_assert(HP <= 32 && !(VAR & (2 | 64 | 128 | 256 | 512)), "the one-launch fit chunk is built for the narrow nets (no streamed weight gradients)");
    const int gtid = PINN_TID, bid = PINN_BID, G = PINN_NBLK;
    const int tid1 = gtid % NTH1, vbid = bid * VW + gtid / NTH1, R = G * VW;       // R partial rows: one per (virtual) workgroup
    const int pc = A0.p_core;
    // VW > 1 (one CU, G = 1): everything an iteration touches lives in LDS behind the virtual workgroups' blocks -- parameters and Adam
    // state, the partial rows, the batch, the pre-pass rows. The first one-CU form kept them in global memory like the grid form and
    // paid ~6 L2 round trips + store fences per iteration (phase clocks: 13.7 us in the tile body for ONE tile, of which 7.3 us its
    // prologue / epilogue; 2.6 us the sum + Adam): as slow as replaying two launches.
    constexpr bool RES = VW > 1;
    using FV = PinnFitVw<HP, ND, N2, MT, LHC, VAR>;
    PINN_SMEM(fit_smem);
    const int pcp = (pc + 3) / 4 * 4;
    // (each pointer from ONE address space, chosen at compile time, so that the compiler can turn the accesses through them into LDS
    //  instructions. Handing them over as OPAQUE generic pointers -- flat accesses the hardware routes to LDS -- was measured: the pass
    //  over one tile 29.8 K -> 34.2 K ticks, slower than the global-memory form. The price of the transparent form: hipcc 7.2 now and then
    //  builds an LDS -> generic -> LDS cast chain whose null test it folds into `v_cmp_ne_u32 0, src_shared_base` -- "Illegal instruction
    //  detected" -- in one kernel of a translation unit or another; pinn_inst.inc leaves those instantiations out)
    float* my;                                             // parameters | exp_avg | exp_avg_sq
    if constexpr (RES) my = fit_smem + FV::BASE_FLOATS; else my = P.state + (size_t)bid * 3 * pc;
    const int sst = RES ? pcp : pc;                        // stride of the three state arrays
    const int dcols = A0.d;
    float* rows_res = my + 3 * pcp;                                // (RES) [VW][pcp]
    float* xs_res = rows_res + VW * pcp;                           // (RES) [n][d]
    float* aux_res = xs_res + (P.n * dcols + 3) / 4 * 4;           // (RES) [n_aux][n]
    for (int i = gtid; i < pc; i += NTH) { my[i] = P.params[i]; my[sst + i] = P.m[i]; my[2 * sst + i] = P.v[i]; }
    const PinnKArgs& A = A0;
#ifndef PINN_EMU
    __shared__ unsigned bail;        // (grid form only)
    if (!RES && gtid == 0) bail = 0u;
#endif
    PINN_FENCE_BLOCK();
    PINN_SYNC();
    const PinnFitCtrl* ctrl = P.ctrl;
    const unsigned k0 = ctrl->k0, k1 = ctrl->k1;
    // -DPINN_FIT_PROF (experiment builds): where an iteration's time goes -- clock ticks of thread 0 per phase, printed at the end
#if defined(PINN_FIT_PROF) && !defined(PINN_EMU)
    long long fp_acc[5] = {0, 0, 0, 0, 0}, fp_last = __builtin_readcyclecounter();
#define FP(i) { const long long fp_now = __builtin_readcyclecounter(); fp_acc[i] += fp_now - fp_last; fp_last = fp_now; }
#else
#define FP(i)
#endif
    // (the trainable-slot mask of this thread's parameters: constant over the chunk -- as a global load inside (d) it put an L2 round
    //  trip in front of every Adam update. One sweep of (d) covers 4 * NTH parameters: every narrow net)
    const bool mask_reg = pc <= 4 * NTH;
    bool upd[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int p = gtid + q * NTH; upd[q] = (p < pc) && (!P.mask || P.mask[p]); }
    const unsigned long long call0 = ctrl->call_index0;
    for (int k = 0; k < P.k_steps; ++k) {
        const float step_size = ctrl->step_size[k], bc2_sqrt = ctrl->bc2_sqrt[k];     // (fetched here: the loads ride behind the tile pass)
        // (a) the batch of iteration k, the points of this (virtual) workgroup's tiles only (the tile body reads nothing else). The
        //     one-CU form draws the batch of iteration k + 1 at the end of (d) instead -- the tile pass is through with the buffer by
        //     then, and the barrier that closes (d) covers it: one barrier and one wait for the stores less per iteration
        float* xs_k;
        if constexpr (RES) xs_k = xs_res; else xs_k = P.xs;
        auto draw = [&](int kk) {
            const unsigned long long call = call0 + (unsigned long long)kk;
            for (long long tile = A.tile_begin + vbid; tile < A.tile_end; tile += R) {
                for (long long i = tile * T + tid1; i < (tile + 1) * T && i < P.n; i += NTH1)
                    pinn_sample_point(xs_k, i, P.spec, k0, k1, (unsigned)(call & 0xffffffffull), (unsigned)(call >> 32));
            }
        };
        if (!RES || k == 0) {
            draw(k);
            PINN_FENCE_BLOCK();
            PINN_SYNC();
        }
        FP(0)
        // (b) forward jets .. reverse sweep of this workgroup's tiles; its partial row(s) of this iteration's parity
        const int rst = RES ? pcp : pc;                    // row stride
        float* rows;
        float* aux_k;
        if constexpr (RES) { rows = rows_res; aux_k = aux_res; } else { rows = P.rows + (size_t)(k & 1) * R * pc; aux_k = A.aux; }
        pinn_tile_body<HP, ND, N2, MT, LHC, ACTC, COMB, VAR, VW>(A, my, rows, xs_k, aux_k, rst);
        FP(1)
        // (c) the rows of this workgroup are complete; one arrive / wait of the grid where there is one
        PINN_FENCE_BLOCK();
        PINN_SYNC();
        FP(2)
#ifndef PINN_EMU
        if (!RES && G > 1) {
            if (gtid == 0) bail = pinn_fit_arrive_and_wait(P.sync, (unsigned)(k + 1) * (unsigned)G);
            __syncthreads();
            if (bail) return;
        }
#endif
        // (d) sum of the partial rows in pinn_reduce_kernel's order (chunks of workgroups c = w mod CH, then the chunks in ascending
        //     order: for R <= CH that is row after row), Adam on this workgroup's own copy
        // (the R row loads of a parameter are independent: issued eight at a time, summed in ascending row order -- the first form of
        //  this loop waited for one L2 round trip per row and made the iteration 2.6x SLOWER than two launches: 43 us against 16)
        for (int p0 = gtid; p0 < pc; p0 += 4 * NTH) {
            double t64[4] = {0.0, 0.0, 0.0, 0.0};      // (double like pinn_reduce_kernel's sums: one rounding per entry)
            for (int c0 = 0; c0 < R; c0 += 8) {
                float r[8][4];
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int p = p0 + q * NTH, c = c0 + j;
                        r[j][q] = (c < R && p < pc) ? rows[(size_t)c * rst + p] : 0.0f;
                    }
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) t64[q] += (double)r[j][q];
            }
            float t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = (float)t64[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = p0 + q * NTH;
                if (p >= pc) continue;
                if (bid == 0) {
                    if (k + 1 == P.k_steps) P.grads[p] = t[q];
                    if (p == P.off_loss) ctrl->loss_base[k] = t[q];
                }
                if (mask_reg ? upd[q] : (!P.mask || P.mask[p])) pinn_adam_update(my, t[q], my + sst, my + 2 * sst, p, step_size, bc2_sqrt, P.b1, P.b2, P.eps);
            }
        }
        if (RES && k + 1 < P.k_steps) draw(k + 1);
        FP(3)
        PINN_FENCE_BLOCK();
        PINN_SYNC();
        FP(4)
    }
#if defined(PINN_FIT_PROF) && !defined(PINN_EMU)
    if (bid == 0 && (gtid == 0 || gtid == NTH - 1) && ctrl->step0 > 1000 && ctrl->step0 < 1200)
        printf("fit prof thread %d k %d: sample+sync %lld  body %lld  sync %lld  reduce+adam %lld  sync %lld ticks\n", gtid, P.k_steps,
               fp_acc[0], fp_acc[1], fp_acc[2], fp_acc[3], fp_acc[4]);
    if (bid == 0 && gtid == 0 && ctrl->step0 > 1000 && ctrl->step0 < 1200) {
        printf("  tile body (cumulative over the launches so far): staging %lld  pre-pass %lld  points+barrier %lld  tile loop %lld  row sums %lld  partial row %lld ticks\n",
               g_pinn_fitprof[0], g_pinn_fitprof[1], g_pinn_fitprof[2], g_pinn_fitprof[3], g_pinn_fitprof[4], g_pinn_fitprof[5]);
    }
#endif
    if (bid == 0) {
        for (int i = gtid; i < pc; i += NTH) { P.params[i] = my[i]; P.m[i] = my[sst + i]; P.v[i] = my[2 * sst + i]; }
        if (RES) for (long long i = gtid; i < P.n * dcols; i += NTH) P.xs[i] = xs_res[i];       // (the last batch, where the eager loop leaves it)
        if (gtid == 0) P.step_ptr[0] = ctrl->step0 + P.k_steps - 1;
    }
}
