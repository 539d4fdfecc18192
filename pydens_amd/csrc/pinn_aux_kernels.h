// pinn_aux_kernels.h -- small non-template kernels (gradient reduction, Adam); included by pinn_abi.cpp only.
#pragma once
#include "pinn_port.h"
#include "pinn_kernel.h"

// ------------------------------------------------------------------------------------------------------------
// x-only pre-pass: evaluates the source terms / variable coefficients of the residual for every point once,
// outside the tile kernel (one thread per point, registers in private memory; N * a-few-ops, microseconds).
// ------------------------------------------------------------------------------------------------------------
PINN_GLOBAL void PINN_LAUNCH_BOUNDS(256)
pinn_aux_kernel(const float* xs, long long n, int d, pinn_program_t pg, float* aux) {
    const long long i = (long long)PINN_BID * 256 + PINN_TID;
    if (i >= n) return;
    float regs[PINN_MAX_REGS];
    for (int c = 0; c < d; ++c) regs[c] = xs[i * d + c];
    for (int k = 0; k < pg.n_ops; ++k) {
        const unsigned w = pg.code[k];
        const int op = w & 255, dst = (w >> 8) & 255, a = (w >> 16) & 255, b = (w >> 24) & 255;
        if (op == PINN_OP_STORE) { aux[(long long)b * n + i] = regs[a]; continue; }
        const float x = (op == PINN_OP_CONST) ? 0.0f : regs[a];
        float y;
        switch (op) {
            case PINN_OP_CONST: y = pg.consts[a]; break;
            case PINN_OP_ADD: y = x + regs[b]; break;
            case PINN_OP_SUB: y = x - regs[b]; break;
            case PINN_OP_MUL: y = x * regs[b]; break;
            case PINN_OP_DIV: y = x / regs[b]; break;
            case PINN_OP_NEG: y = -x; break;
            case PINN_OP_SIN: y = sinf(x); break;
            case PINN_OP_COS: y = cosf(x); break;
            case PINN_OP_EXP: y = expf(x); break;
            case PINN_OP_LOG: y = logf(x); break;
            case PINN_OP_TANH: y = tanhf(x); break;
            case PINN_OP_SQRT: y = sqrtf(x); break;
            case PINN_OP_POW: y = powf(x, pg.consts[b]); break;
            case PINN_OP_ABS: y = fabsf(x); break;
            case PINN_OP_SIGMOID: y = 1.0f / (1.0f + expf(-x)); break;
            case PINN_OP_RECIP: y = 1.0f / x; break;
            default: y = x; break;
        }
        regs[dst] = y;
    }
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

PINN_DEVICE void pinn_adam_update(float* params, float gi, float* m, float* v, long long i, float step_size, float bc2_sqrt,
                                  float b1, float b2, float eps) {
    const float mi = m[i] + (1.0f - b1) * (gi - m[i]);       // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = fmaf(1.0f - b2, gi * gi, b2 * v[i]);    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    params[i] -= step_size * (mi / denom);
}

#ifndef PINN_REDUCE_PB
#define PINN_REDUCE_PB 32
#endif
PINN_GLOBAL void PINN_LAUNCH_BOUNDS(1024)
pinn_reduce_kernel(const float* partials, int n_wg, int p_core, float* grads, int accumulate, int do_adam, float* params,
                   float* m, float* v, const unsigned char* mask, int step_value, float step_size, float bc2_sqrt, float b1,
                   float b2, float eps, int* step_ptr, float* loss_out, int off_loss) {
    PINN_SMEM(red);
    const int tid = PINN_TID;
    constexpr int PB = PINN_REDUCE_PB, CH = 1024 / PB;        // PB parameters x CH chunks of workgroups per block
    const int pl = tid % PB, ch = tid / PB;
    const int p = PINN_BID * PB + pl;
    float s = 0.0f;
    if (p < p_core)
        for (int w = ch; w < n_wg; w += CH) s += partials[(size_t)w * p_core + p];
    red[ch * PB + pl] = s;
    PINN_SYNC();
    if (tid < PB && p < p_core) {
        float t = 0.0f;
        for (int c = 0; c < CH; ++c) t += red[c * PB + tid];
        if (accumulate) t += grads[p];
        grads[p] = t;
        if (loss_out && p == off_loss) loss_out[0] = t;
        if (do_adam && (!mask || mask[p])) pinn_adam_update(params, t, m, v, p, step_size, bc2_sqrt, b1, b2, eps);
    }
    if (do_adam && PINN_BID == 0 && tid == 0) step_ptr[0] = step_value;
}

// ------------------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam single-tensor form, model_torch.py:461): exp_avg/exp_avg_sq EMA, bias-corrected step
// ------------------------------------------------------------------------------------------------------------
// wt[l][in][out] = W_l[out][in] for the lh hidden->hidden matrices (hp x hp, row stride hp, layer stride hidden_stride):
// 32 x 32 tiles through LDS, both sides coalesced. Grid: (hp/32)^2 * lh workgroups of 256 threads.
PINN_GLOBAL void PINN_LAUNCH_BOUNDS(256) pinn_transpose_kernel(const float* wh, int hidden_stride, int hp, float* wt) {
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

PINN_GLOBAL void PINN_LAUNCH_BOUNDS(64) pinn_tick_kernel(int* step_ptr) {
    if (PINN_TID == 0 && PINN_BID == 0) step_ptr[0] += 1;
}

PINN_GLOBAL void PINN_LAUNCH_BOUNDS(256)
pinn_adam_kernel(float* params, const float* grads, float* m, float* v, const unsigned char* mask, long long n,
                 int* step_ptr, int step_value, float lr, float step_size, float bc2_sqrt, float b1, float b2, float eps) {
    // step_value > 0: the host counts (step_size / bc2_sqrt come with it, the count is mirrored to step_ptr); otherwise
    // the count lives on the device and the bias corrections are computed here
    const long long i = (long long)PINN_BID * 256 + PINN_TID;
    if (step_value > 0 && i == 0) step_ptr[0] = step_value;
    if (i >= n) return;
    if (mask && !mask[i]) return;
    if (step_value <= 0) pinn_adam_scalars((double)step_ptr[0], lr, b1, b2, &step_size, &bc2_sqrt);
    pinn_adam_update(params, grads[i], m, v, i, step_size, bc2_sqrt, b1, b2, eps);
}
