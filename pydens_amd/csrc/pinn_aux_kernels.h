// pinn_aux_kernels.h -- small non-template kernels (gradient reduction, Adam); included by pinn_abi.cpp only.
#pragma once
// (the small kernels of this header are launched from pinn_abi.cpp only; the per-width units of widths <= 32 include the header for the
//  structs and device functions pinn_fit_kernel.h shares with them -- there the kernels get internal linkage and are dropped unused)
#ifdef PINN_AUX_KERNELS_STATIC
#define PINN_AUX_GLOBAL static PINN_GLOBAL
#else
#define PINN_AUX_GLOBAL PINN_GLOBAL
#endif

#include "pinn_port.h"
#include "pinn_kernel.h"

// ------------------------------------------------------------------------------------------------------------
// x-only pre-pass: evaluates the source terms / variable coefficients of the residual for every point once,
// outside the tile kernel (one thread per point, registers in private memory; N * a-few-ops, microseconds).
// ------------------------------------------------------------------------------------------------------------
PINN_AUX_GLOBAL void PINN_LAUNCH_BOUNDS(256)
pinn_aux_kernel(const float* xs, long long n, int d, pinn_program_t pg, PinnPreConsts c64, float* aux) {
    const long long i = (long long)PINN_BID * 256 + PINN_TID;
    if (i >= n) return;
    double regs[PINN_MAX_REGS];                 // private (scratch): the program indexes it at run time; fp64 like the in-kernel form
    pinn_prepass_point(pg, c64, xs + i * d, d, aux, n, i, regs, 1);
}

// ------------------------------------------------------------------------------------------------------------
// sum of the per-workgroup partial gradients (fixed order => deterministic): block = 64 parameters x 16 chunks of
// workgroups (every wave reads whole 256-B rows), LDS tree over the chunks; optionally the Adam update of those 64
// parameters right behind it (single-rank steps: no all-reduce in between, two launches less).
// ------------------------------------------------------------------------------------------------------------
// bias corrections in double like torch's Python-side scalars (1 - beta ** step); the host computes them when it knows
// the step (two double pow per thread in the tail of every block otherwise)
PINN_HOST_DEVICE inline void pinn_adam_scalars(double t, float lr, float b1, float b2, float* step_size, float* bc2_sqrt) {
    const double bc1 = 1.0 - pow((double)b1, t);
    const double bc2 = 1.0 - pow((double)b2, t);
    *step_size = (float)((double)lr / bc1);
    *bc2_sqrt = (float)sqrt(bc2);
}

// (the arithmetic of one Adam update on operands already in registers: ONE expression tree for every caller, so that every path
//  rounds -- and contracts multiply-adds -- the same way)
PINN_DEVICE void pinn_adam_apply(float* params, float* m, float* v, long long i, float gi, float m_old, float v_old, float p_old,
                                 float step_size, float bc2_sqrt, float b1, float b2, float eps) {
    const float mi = m_old + (1.0f - b1) * (gi - m_old);     // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = fmaf(1.0f - b2, gi * gi, b2 * v_old);   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    params[i] = p_old - step_size * (mi / denom);
}

PINN_DEVICE void pinn_adam_update(float* params, float gi, float* m, float* v, long long i, float step_size, float bc2_sqrt,
                                  float b1, float b2, float eps) {
    pinn_adam_apply(params, m, v, i, gi, m[i], v[i], params[i], step_size, bc2_sqrt, b1, b2, eps);
}

#ifndef PINN_REDUCE_PB
#define PINN_REDUCE_PB 32
#endif
// ------------------------------------------------------------------------------------------------------------
// Collocation sampler on the device: replaces the host-side draws of reference model_torch.py:430-434 (`torch.rand`
// per input column, or `sampler.sample(batch_size)` of a batchflow NumpySampler product `a & b & ...`).
// Counter-based generator Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
// SC'11; constants of the Random123 library): no state in memory, one launch per batch whatever the number of columns,
// every (seed, call, point, column) addresses its own random word, so a batch does not depend on the launch geometry.
// oracle/philox.py restates it in numpy (pinned to the Random123 known-answer vectors); tests compare bit for bit.
// ------------------------------------------------------------------------------------------------------------
struct PinnSampleSpec {
    int d;
    int kind[PINN_MAX_INPUTS];      // PINN_SAMPLE_UNIFORM / _NORMAL / _CONST
    float a[PINN_MAX_INPUTS], b[PINN_MAX_INPUTS];
};

PINN_DEVICE void pinn_philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                    unsigned (&out)[4]) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// products / sums that must round like the numpy oracle: separately, never contracted into a fused multiply-add
// (hipcc's __fmul_rn / __fadd_rn are plain operators and do get contracted, hence the pragma in the function bodies)
PINN_DEVICE float pinn_mul_then_add(float a, float w, float u) {
#pragma clang fp contract(off)
    const float t = w * u;
    return a + t;
}

// the d columns of point i of batch `call` (what one thread of pinn_sample_kernel does; also the tail of pinn_reduce_kernel when a fit
// chunk lets the reduction of iteration k draw the batch of iteration k + 1: one launch less per iteration)
PINN_DEVICE void pinn_sample_point(float* xs, long long i, const PinnSampleSpec& spec, unsigned k0, unsigned k1, unsigned call_lo,
                                   unsigned call_hi) {
    const unsigned i_lo = (unsigned)((unsigned long long)i & 0xffffffffull), i_hi = (unsigned)((unsigned long long)i >> 32);
    unsigned r[4] = {0u, 0u, 0u, 0u};
    for (int c = 0; c < spec.d; ++c) {
        if ((c & 3) == 0) pinn_philox4x32_10(i_lo, i_hi, call_lo, (call_hi & 0x0fffffffu) | ((unsigned)(c >> 2) << 28), k0, k1, r);
        const float a = spec.a[c], b = spec.b[c];
        float v = a;
        if (spec.kind[c] == PINN_SAMPLE_UNIFORM) {
            const float u = (float)(r[c & 3] >> 8) * 5.9604644775390625e-8f;              // 24 bits -> [0, 1)
            v = pinn_mul_then_add(a, b - a, u);
        } else if (spec.kind[c] == PINN_SAMPLE_NORMAL) {
            unsigned q[4];
            pinn_philox4x32_10(i_lo, i_hi, call_lo, (call_hi & 0x0fffffffu) | ((unsigned)(8 + c) << 28), k0, k1, q);
            const float u1 = (float)((q[0] >> 8) + 1u) * 5.9604644775390625e-8f;          // (0, 1]
            const float u2 = (float)(q[1] >> 8) * 5.9604644775390625e-8f;
            const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
            v = pinn_mul_then_add(a, b, z);
        }
        xs[i * spec.d + c] = v;
    }
}
// what the reduction of a fit iteration needs to draw the NEXT iteration's batch (n == 0: nothing to draw)
struct PinnNextBatch {
    float* xs; long long n; PinnSampleSpec spec; unsigned k0, k1; unsigned long long call;
};

// Replayable launch graph of a chunk of fit iterations (pinn_fit_steps, batches of a few thousand points: the latency regime): what
// changes from one iteration to the next -- the Philox batch counter, the Adam step with its bias corrections, the slot of the loss
// history -- is read from this block in device memory, indexed by the iteration number k that is baked into the graph's kernel nodes;
// the block itself is rewritten (pinn_fit_ctrl_kernel, an ordinary launch in front of the graph) for every chunk.
#define PINN_FIT_CHUNK_MAX 128
struct PinnFitCtrl {
    unsigned long long call_index0;     // Philox batch counter of iteration 0 of the chunk
    float* loss_base;                   // entry 0 of the chunk in the loss history
    int step0, pad;                     // Adam step number of iteration 0
    unsigned k0, k1;                    // Philox key of the fit call's sampler (round 5: the key changes per fit call; in the control
                                        // block, not baked into the graph's nodes, a recorded chunk survives across fit calls)
    float step_size[PINN_FIT_CHUNK_MAX], bc2_sqrt[PINN_FIT_CHUNK_MAX];      // pinn_adam_scalars per iteration (computed on the host in
                                                                             // double, as for the eager loop: bit-identical updates)
};
struct PinnFitCtrlArgs { PinnFitCtrl c; };
PINN_AUX_GLOBAL void PINN_LAUNCH_BOUNDS(128) pinn_fit_ctrl_kernel(PinnFitCtrl* dst, PinnFitCtrlArgs a) {
    const int t = PINN_TID;
    if (t == 0) { dst->call_index0 = a.c.call_index0; dst->loss_base = a.c.loss_base; dst->step0 = a.c.step0; dst->pad = 0;
                  dst->k0 = a.c.k0; dst->k1 = a.c.k1; }
    if (t < PINN_FIT_CHUNK_MAX) { dst->step_size[t] = a.c.step_size[t]; dst->bc2_sqrt[t] = a.c.bc2_sqrt[t]; }
}

PINN_AUX_GLOBAL void PINN_LAUNCH_BOUNDS(1024)
pinn_reduce_kernel(const float* partials, int n_wg, int p_core, float* grads, int accumulate, int do_adam, float* params,
                   float* m, float* v, const unsigned char* mask, int step_value, float step_size, float bc2_sqrt, float b1,
                   float b2, float eps, int* step_ptr, float* loss_out, int off_loss, const PinnFitCtrl* ctrl, int ctrl_k,
                   PinnNextBatch next) {
    PINN_SMEM(red);
    const int tid = PINN_TID;
    if (ctrl) {             // (graph replay: this iteration's Adam step and loss slot come from the control block)
        step_value = ctrl->step0 + ctrl_k;
        step_size = ctrl->step_size[ctrl_k];
        bc2_sqrt = ctrl->bc2_sqrt[ctrl_k];
        loss_out = ctrl->loss_base + ctrl_k;
    }
    constexpr int PB = PINN_REDUCE_PB, CH = 1024 / PB;        // PB parameters x CH chunks of workgroups per block
    const int pl = tid % PB, ch = tid / PB;
    const int p = PINN_BID * PB + pl;
    // the operands of the final lanes' Adam update do not depend on the sums: fetched up front, so that their round trip runs under the
    // row loads instead of behind the barrier (round 5: this kernel took 8 us on BASELINE config 2 -- 4 % of the step -- for 13 MB;
    // it was a chain of dependent round trips: row after row, then the mask, then m / v / the parameter)
    const bool fin = tid < PB && p < p_core;
    bool upd = false;
    float g_old = 0.0f, m_old = 0.0f, v_old = 0.0f, p_old = 0.0f;
    if (fin) {
        if (accumulate) g_old = grads[p];
        if (do_adam) {
            upd = !mask || mask[p];
            if (upd) { m_old = m[p]; v_old = v[p]; p_old = params[p]; }
        }
    }
    // the rows of this thread's chunk, eight loads in flight at a time, summed in ascending row order (as before) -- in DOUBLE since
    // round 6: a gradient entry that is a cancelling sum over the batch (BASELINE config 4's d loss / d b_L: sum of |terms| 850 x the
    // result) walks through prefix sums far larger than its total, and every fp32 add of this loop rounded at THEIR magnitude: 5e-6
    // relative on that entry from the 256 rows alone (tools/cfg4_bl_probe.py). One rounding to fp32 at the end instead; the adds are
    // free beside the row loads.
    double s = 0.0;
    if (p < p_core) {
        for (int w0 = ch; w0 < n_wg; w0 += 8 * CH) {
            float r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int w = w0 + j * CH;
                r[j] = (w < n_wg) ? partials[(size_t)w * p_core + p] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (w0 + j * CH < n_wg) s += (double)r[j];
        }
    }
    double* red64 = reinterpret_cast<double*>(red);
    red64[ch * PB + pl] = s;
    PINN_SYNC();
    if (fin) {
        double t64 = 0.0;
        for (int c = 0; c < CH; ++c) t64 += red64[c * PB + tid];
        if (accumulate) t64 += (double)g_old;
        const float t = (float)t64;
        grads[p] = t;
        if (loss_out && p == off_loss) loss_out[0] = t;
        if (upd) pinn_adam_apply(params, m, v, p, t, m_old, v_old, p_old, step_size, bc2_sqrt, b1, b2, eps);
    }
    if (do_adam && PINN_BID == 0 && tid == 0) step_ptr[0] = step_value;
    // fit chunks: this iteration's tile kernel is through with the batch buffer -- the batch of the next iteration is drawn here
    // (same generator, same counters as pinn_sample_kernel: bit-identical batches), which saves the iteration a dependent launch
    if (next.n > 0) {
        unsigned long long call = next.call;
        unsigned nk0 = next.k0, nk1 = next.k1;
        if (ctrl) { call = ctrl->call_index0 + (unsigned long long)(ctrl_k + 1); nk0 = ctrl->k0; nk1 = ctrl->k1; }
        for (long long i = (long long)PINN_BID * 1024 + tid; i < next.n; i += (long long)PINN_NBLK * 1024)
            pinn_sample_point(next.xs, i, next.spec, nk0, nk1, (unsigned)(call & 0xffffffffull), (unsigned)(call >> 32));
    }
}

// ------------------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam single-tensor form, model_torch.py:461): exp_avg/exp_avg_sq EMA, bias-corrected step
// ------------------------------------------------------------------------------------------------------------
// wt[l][in][out] = W_l[out][in] for the lh hidden->hidden matrices (hp x hp, row stride hp, layer stride hidden_stride):
// 32 x 32 tiles through LDS, both sides coalesced. Grid: (hp/32)^2 * lh workgroups of 256 threads.
PINN_AUX_GLOBAL void PINN_LAUNCH_BOUNDS(256) pinn_transpose_kernel(const float* wh, int hidden_stride, int hp, float* wt) {
    PINN_SMEM(tile);                                     // [32][33]
    const int tiles = hp / 32;
    const int l = PINN_BID / (tiles * tiles), t = PINN_BID % (tiles * tiles), tr = t / tiles, tc = t % tiles;
    const float* src = wh + (size_t)l * hidden_stride;
    float* dst = wt + (size_t)l * hp * hp;
    const int x = PINN_TID & 31, y0 = PINN_TID >> 5;
    for (int y = y0; y < 32; y += 8) tile[y * 33 + x] = src[(size_t)(tr * 32 + y) * hp + tc * 32 + x];
    PINN_SYNC();
    for (int y = y0; y < 32; y += 8) dst[(size_t)(tc * 32 + y) * hp + tr * 32 + x] = tile[x * 33 + y];
}

// ------------------------------------------------------------------------------------------------------------
// split-bf16 kernels (pinn_tile_kernel VAR 512): the hidden->hidden weights as MFMA A-operand fragments of three bf16 planes,
// w = hi + mid + lo exactly (pinn_split4's arithmetic), in the order the waves load them -- PinnCfg::wsp_frag:
// [layer][direction][K block of 32][16-unit tile j][plane][lane] x 16 bytes. Direction 0 (forward GEMM): lane (lr, lq) holds
// W[16 j + lr][32 kb + 8 lq + e], e = 0..7; direction 1 (data gradient): W[32 kb + 8 lq + e][16 j + lr]. One thread per
// (layer, direction, K block, tile, lane); rewritten before every step (the weights change every step), 147 KB at 3 x 64 x 64.
// ------------------------------------------------------------------------------------------------------------
PINN_AUX_GLOBAL void PINN_LAUNCH_BOUNDS(256) pinn_wsplit_kernel(const float* wh, int hidden_stride, int hp, int lh, pinn_s16x8* out) {
    const int kbs = hp / 32, nt = hp / 16;
    const int idx = PINN_BID * 256 + PINN_TID;
    if (idx >= lh * 2 * kbs * nt * 64) return;
    const int lane = idx & 63, j = (idx >> 6) % nt, kb = (idx >> 6) / nt % kbs, dir = (idx >> 6) / nt / kbs % 2, l = (idx >> 6) / nt / kbs / 2;
    const int lr = lane & 15, lq = lane >> 4;
    const float* W = wh + (size_t)l * hidden_stride;
    unsigned b[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int unit = 16 * j + lr, k = 32 * kb + 8 * lq + e;
        const float x = dir == 0 ? W[(size_t)unit * hp + k] : W[(size_t)k * hp + unit];
        pinn_split3(x, b[0][e], b[1][e], b[2][e]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        pinn_s16x8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (short)(b[p][e] >> 16);
        out[((((size_t)(l * 2 + dir) * kbs + kb) * nt + j) * 3 + p) * 64 + lane] = f;
    }
}

PINN_AUX_GLOBAL void PINN_LAUNCH_BOUNDS(64) pinn_tick_kernel(int* step_ptr) {
    if (PINN_TID == 0 && PINN_BID == 0) step_ptr[0] += 1;
}

PINN_AUX_GLOBAL void PINN_LAUNCH_BOUNDS(256)
pinn_adam_kernel(float* params, const float* grads, float* m, float* v, const unsigned char* mask, long long n,
                 int* step_ptr, int step_value, float lr, float step_size, float bc2_sqrt, float b1, float b2, float eps,
                 float* loss_out, int off_loss) {
    // step_value > 0: the host counts (step_size / bc2_sqrt come with it, the count is mirrored to step_ptr); otherwise
    // the count lives on the device and the bias corrections are computed here
    // loss_out: the loss slot of the (all-reduced) gradient buffer is copied to one more address -- entry i of the host's
    // loss history (model_torch.py:464) -- so that recording an iteration costs no launch of its own
    const long long i = (long long)PINN_BID * 256 + PINN_TID;
    if (step_value > 0 && i == 0) step_ptr[0] = step_value;
    if (loss_out && i == 0) loss_out[0] = grads[off_loss];
    if (i >= n) return;
    if (mask && !mask[i]) return;
    if (step_value <= 0) pinn_adam_scalars((double)step_ptr[0], lr, b1, b2, &step_size, &bc2_sqrt);
    pinn_adam_update(params, grads[i], m, v, i, step_size, bc2_sqrt, b1, b2, eps);
}

// one thread per point. Counter = (point low, point high, call low, call high | block << 28): block b < 8 supplies the
// uniform words of columns 4b .. 4b+3, block 8 + c the two extra words of a normal column c (Box-Muller).
PINN_AUX_GLOBAL void PINN_LAUNCH_BOUNDS(256)
pinn_sample_kernel(float* xs, long long n, PinnSampleSpec spec, unsigned k0, unsigned k1, unsigned call_lo, unsigned call_hi,
                   const PinnFitCtrl* ctrl, int ctrl_k) {
    const long long i = (long long)PINN_BID * 256 + PINN_TID;
    if (i >= n) return;
    if (ctrl) {             // (graph replay: the batch counter of this iteration)
        const unsigned long long call = ctrl->call_index0 + (unsigned long long)ctrl_k;
        call_lo = (unsigned)(call & 0xffffffffull); call_hi = (unsigned)(call >> 32);
        k0 = ctrl->k0; k1 = ctrl->k1;
    }
    pinn_sample_point(xs, i, spec, k0, k1, call_lo, call_hi);
}
