// pinn_kernel.h -- the fused PINN tile kernel (forward Taylor-mode jets -> ansatz -> residual -> reverse sweep).
//
// Replaces, for one tile of T collocation points per workgroup pass, the whole per-iteration arithmetic of the
// reference `Solver.fit` (pydens/model_torch.py:437-460): MLP forward (:170-172), the nested-autograd sweeps of
// D(...) (:174-178), the ansatz (:107-128), the user equation + MSELoss (:447-448) and loss.backward() (:460).
//
// Execution model (gfx950): one workgroup = NW waves; wave w owns output units [16*NTW*w, 16*NTW*(w+1)) of every
// hidden layer.  The S derivative streams of a tile (value, d/dx_k, d2/dx_k2) are stacked as extra GEMM rows and
// pushed through each fc layer with v_mfma_f32_16x16x4_f32 (exact fp32): A = activations from LDS
// (ds_read_b128 along K), B = weight fragment held in VGPRs for all S*MT row tiles.  The activation jets are
// evaluated on the MFMA accumulators in registers (all S streams of one (point, unit) live in one lane).
// Weight gradients are accumulated in MFMA accumulators that persist across all tiles of the (persistent)
// workgroup; per-workgroup partials are summed by pinn_reduce_kernel.  See DESIGN.md sections 3-5.
//
// Build knobs (all have the product's value as default; `tools/variant.sh <name> -D...` builds an A/B library, the
// measurements behind each default are in DESIGN.md section 6; knobs whose experiments were closed as negative are gone,
// their findings stay as comments where the code was):
//   PINN_LDA_PAD (8)              padding of the LDS activation rows (bank spread)
//   PINN_WTG_MIN_HP (128) / PINN_ONEBUF_MIN_HP (256) / PINN_SLAB_NT_MIN_HP (256) / PINN_GZ_LATE_MAX_HP (128) / PINN_WGX_MIN_HP
//                                 structure thresholds by width: W^T from a global copy, one LDS buffer, non-temporal slabs,
//                                 gz stores behind the GEMM, streamed weight-gradient kernel (pinn_inst.inc)
//   PINN_CFG2_SLABL (0)           64: saved jets of the cfg2 kernel in LDS instead of the global slab (pinn_inst.inc)
//   PINN_SVPF_MAX (8), PINN_REGB_MAX (6), PINN_REGB_MAX_SPEC (8), PINN_WAVES_PER_SIMD (1), PINN_SCHED_IL (1, pinn_port.h)
//                                 prefetch / register-accumulator / occupancy / scheduling thresholds
//   PINN_FAST_MT_S5 / _S2 / _COMB, PINN_FAST_VAR_COMB, PINN_WIDE_MT (pinn_inst.inc)   tile heights of the fast kernels
//   PINN_SP_ROUND (2), PINN_SP_PIPE / PINN_SP_PIPE_W (per translation unit: build.py)   split-bf16 kernels
//   PINN_TEAM_FLAGS (0)           team-local LDS arrival counters instead of s_barrier in the two-team kernels (measured slower)
//   PINN_PTALL (round 5)          shape-specialised kernels: the point stage (ansatz, residual, their reverse) evaluated by EVERY lane for
//                                 its own point (16 * MT points, replicated over the unit quads and the waves) instead of by the first T
//                                 threads: the LDS round trip of the upstream gradient and the barrier behind the point stage disappear
//   PINN_TANH_POLY (round 5)      tanh of small arguments by an odd minimax polynomial (relative accuracy where e^{-2|z|} cancels)
//   PINN_ONLY_BASELINE            experiment builds: only the BASELINE kernels (seconds to compile)
//   PINN_DEBUG_ABI                experiment builds: pinn_debug_set_flags / pinn_debug_phase_buffer and the kernel paths behind them
//   PINN_PROFILE_PHASES           per-phase cycle counters (tools/phases.py);  PINN_ABL  timing ablations
#pragma once
#include "pinn_port.h"

#include "../../include/pinn.h"

#ifndef PINN_EMU
#define PINN_HOST_DEVICE __host__ __device__
#else
#define PINN_HOST_DEVICE
#endif

enum { PINN_MODE_FORWARD = 0, PINN_MODE_STEP = 1, PINN_MODE_BACKWARD = 2 };
#ifndef PINN_PTALL
#define PINN_PTALL 1            // (documented at its use in the tile kernel)
#endif
#define PINN_PTALL_DEFAULT PINN_PTALL

// activation code of activation index a (4 bits each, 16 per word)
PINN_HOST_DEVICE inline int pinn_act_code(const unsigned long long (&codes)[2], int a) {
    return (int)((codes[a >> 4] >> (4 * (a & 15))) & 15ull);
}

// an activation as the jet functions see it: its code (PINN_ACT_*) and the ONE parameter some of torch's modules carry -- round 6:
// nn.LeakyReLU(negative_slope), nn.ELU(alpha), nn.Softplus(beta) configured away from their defaults (the reference hands module
// instances through, model_torch.py:150; pinn_set_act_params). Converts to and from the plain code, so code that only compares
// codes is written as before; a kernel whose activation is a compile-time constant carries no parameter at all.
struct PinnAct {
    int c; float p;
    PINN_HOST_DEVICE PinnAct(int code = 0, float par = 0.0f) : c(code), p(par) {}
    PINN_HOST_DEVICE operator int() const { return c; }
};

// timing-experiment bits (kernels skip loads / stores / barriers; include/pinn.h): compiled into -DPINN_DEBUG_ABI builds only
// (tools/variant.sh), the product kernels carry none of these paths
#ifdef PINN_DEBUG_ABI
#define PINN_DBG(A, bit) (((A).debug_flags & (bit)) != 0)
#else
#define PINN_DBG(A, bit) false
#endif

// Higher-order stream counts travel PACKED in one integer wherever the interface says "n2" (template parameter N2P, the
// n2 argument of the C-ABI): low three bits = directions with a second derivative, bits 3.. = how many of THOSE (the
// first ones) also carry a third derivative, bits 6.. how many of those a fourth. Stream layout: [u | firsts (nd) | seconds (n2) | thirds (n3) | fourths (n4)].
PINN_HOST_DEVICE constexpr int pinn_n2(int n2p) { return n2p & 7; }
PINN_HOST_DEVICE constexpr int pinn_n3(int n2p) { return (n2p >> 3) & 7; }
PINN_HOST_DEVICE constexpr int pinn_n4(int n2p) { return n2p >> 6; }      // round 5: of THOSE (the first ones), how many also carry a fourth derivative
PINN_HOST_DEVICE constexpr int pinn_ns(int nd, int n2p) { return 1 + nd + (n2p & 7) + ((n2p >> 3) & 7) + (n2p >> 6); }

constexpr int PINN_LHMAX = 4;      // hidden->hidden layers whose dW accumulators live in registers
constexpr int PINN_XS_LD = PINN_MAX_INPUTS;

// constants of the x-only pre-pass in double precision (pinn_residual_t::pre_consts64)
struct PinnPreConsts { double v[PINN_MAX_CONSTS]; };

struct PinnKArgs {
    const float* params;
    const float* xs;             // [N][d]
    const float* ic_streams;     // [S_user][N] or null
    const float* gin;            // MODE_BACKWARD: [S_user][N]
    float* out_streams;          // MODE_FORWARD: [S_user][N]
    float* partials;             // [nWG][p_core]
    f32x4* slab;                 // saved activations, lane private (per workgroup; per TILE of the launch for WGX kernels)
    f32x4* gzslab;               // WGX kernels: pre-activation gradients gz_a of every tile, the A operand of pinn_wgrad_kernel
    long long tile_begin, tile_end;   // tiles [tile_begin, tile_end) of the batch belong to this launch (WGX launches go chunk by chunk)
    int partial_row0;            // pinn_wgrad_kernel: partial rows >= this one belong to no tile-kernel workgroup (zeroed there)
    const float* wt;             // widths >= 128: transposed copy of the hidden weights, [lh][in][out] (pinn_transpose_kernel)
    const void* wsp;             // split-bf16 kernels (VAR 512): hidden weights as hi / mid / lo bf16 MFMA fragments (pinn_wsplit_kernel)
    int gemm_mode;               // PINN_GEMM_FP32 / PINN_GEMM_BF16X3 (pinn_set_gemm_mode): which instantiation the launcher picks
    int tanh_mode;               // PINN_TANH_FAST / PINN_TANH_ACCURATE (pinn_set_tanh_mode): likewise
    long long* prof;             // optional per-phase cycle counters (PINN_PROFILE_PHASES builds only)
    int debug_flags;             // timing-experiment bits (PINN_DBG; -DPINN_DEBUG_ABI builds only)
    long long n_points;
    int lh, d, act, mode;        // act: the activation code shared by every layer, or -1 when they differ (act_codes)
    unsigned long long act_codes[2];   // 4 bits per activation index a = 0..lh (a = 0: first layer): pinn_act_code
    float act_par[PINN_MAX_LAYERS];    // parameter of activation a (LeakyReLU negative_slope / ELU alpha / Softplus beta; pinn_set_act_params)
    int n_skips;                 // skip connections 'R ... +': h_out[skip_dst] += h_out[skip_src] (activation indices)
    int skip_src[PINN_MAX_SKIPS], skip_dst[PINN_MAX_SKIPS];
    int skip_pre;                // bit k: skip k ends IN FRONT of the activation ('R fa f+ a': z[skip_dst] += h_out[skip_src])
    int skip_src_pre;            // bit k: skip k STARTS in front of the activation ('f R a ...': the pre-activation jets are carried)
    int skip_outer;              // bit k: another skip opens while skip k is open (nested 'R .. R .. + .. +', round 5): the jets skip k carries
                                 // travel through its slab slot instead of the one register set (full breadth kernels, generic depth)
    int off_b1, off_wh, hidden_stride, off_wl, off_bl, off_ls, off_loss;
    int p_core;                  // row stride of `partials` = length of the gradient buffer (user slots included)
    int off_extra, n_vars;       // first user slot; V(...) scalars a residual program reads (registers S+d+n_aux+k)
    int ic_var1;                 // 1 + user slot holding a trainable constant initial value (0: the ic_const argument)
    int ic_rows;                 // 1: IC streams = pre-pass rows ic_row[s] (>= 0) or constants ic_cst[s]
    int ic_row[PINN_MAX_STREAMS];
    float ic_cst[PINN_MAX_STREAMS];
    int ndims, nsp, has_bc, has_ic;
    float bc_value, t0, ic_const, inv_n;
    float lo[PINN_MAX_INPUTS], hi[PINN_MAX_INPUTS];
    int dir_cols[PINN_MAX_DIRS];
    int s_user;                  // streams visible to the caller (<= S of the instantiation)
    int comb;                    // 1: the single second-order stream is the combination sum_k comb_w[k] d2/dx_k2
    float inv_w[PINN_MAX_INPUTS];    // 1 / (hi - lo)
    float comb_w[PINN_MAX_DIRS];     // COMB instantiations: weights c_k of the combined second-order stream sum_k c_k d2/dx_k2
    // residual (MODE_STEP)
    int res_kind, n_aux, src_row;
    float src_const;
    float coef[PINN_MAX_STREAMS];
    int coef_row[PINN_MAX_STREAMS];
    float* aux;                  // [n_aux][N] rows of the x-only pre-pass
    pinn_program_t prog;
    int pre_nregs;               // registers the pre-pass program touches (inputs included)
    pinn_program_t pre;          // n_ops > 0: the tile kernel evaluates the pre-pass itself for the points of its own tiles
                                 // (kernel prologue) instead of a separate launch in front of it
    PinnPreConsts pre_consts64;  // ... in fp64, with these constants (round 6)
};

template <int HP_, int ND_, int N2_, int MT_ = 1, bool SPLIT_ = false>
struct PinnCfg {
    static constexpr int HP = HP_, ND = ND_, N2 = pinn_n2(N2_), N3 = pinn_n3(N2_), MT = MT_;
    static constexpr bool SPLIT = SPLIT_;                    // split-bf16 GEMM operands (VAR 512; below)
    static constexpr int S = pinn_ns(ND_, N2_);
    static constexpr int NT = HP / 16;                       // 16-wide unit tiles
    static constexpr int NW = (NT <= 4) ? NT : 8;            // waves per workgroup (four waves with twice the unit tiles at widths
                                                             // >= 128: one wave per SIMD, 12-20 % slower -- DESIGN.md section 6a)
    static constexpr int NTW = NT / NW;                      // unit tiles per wave
    static constexpr int T = 16 * MT;                        // points per tile
#ifndef PINN_LDA_PAD
#define PINN_LDA_PAD 8
#endif
    static constexpr int LDA = HP + PINN_LDA_PAD;            // row stride of point-major activation buffers (bank spread)
    static constexpr int NTHREADS = NW * 64;
    // widths above 128: ONE activation buffer (two would not fit the 160 KB of LDS): the forward pass works in place and
    // the reverse pass keeps the weight-gradient B fragments of h_{a-1} in registers while gz_a replaces it in LDS
#ifndef PINN_ONEBUF_MIN_HP
#define PINN_ONEBUF_MIN_HP 256
#endif
    static constexpr bool ONEBUF = HP >= PINN_ONEBUF_MIN_HP;
#ifndef PINN_WTG_MIN_HP
#define PINN_WTG_MIN_HP 128                                  // (experiment builds lower it: W^T from global memory instead of LDS)
#endif
    static constexpr bool WTG = HP >= PINN_WTG_MIN_HP;       // data-gradient GEMM reads a transposed weight copy (A.wt)
    // LDS carve (floats); every offset is a multiple of 4 floats (16 B, ds_read_b128 alignment)
    static constexpr int O_XS = 0;
    static constexpr int O_W1 = O_XS + 2 * T * PINN_XS_LD;      // points of a tile, double-buffered
    static constexpr int O_B1 = O_W1 + HP * PINN_XS_LD;
    static constexpr int O_WL = O_B1 + HP;
    static constexpr int O_BUFA = O_WL + HP;
    // split-bf16 kernels keep an activation buffer as THREE bf16 planes (hi, mid, lo) of [S*T rows][HP units]: rows of
    // 2 HP bytes without padding, the 16-byte chunk index XORed with (row & 7) instead (tools/layout/lds_banks.py:
    // ds_read_b128 along K and ds_read_b64_tr_b16 along the points are conflict-free, ds_write_b64 two-way)
    static constexpr int SP_ROW_BYTES = 2 * HP;
    static constexpr int SP_PLANE_BYTES = S * T * SP_ROW_BYTES;
    static constexpr int BUF_FLOATS = SPLIT ? 3 * SP_PLANE_BYTES / 4 : S * T * LDA;
    static constexpr int O_BUFB = ONEBUF ? O_BUFA : O_BUFA + BUF_FLOATS;
    static constexpr int O_NET = O_BUFB + BUF_FLOATS;           // [NW][S][T] per-wave partial dot products
    // floats of the activation buffers per point of a tile: where the in-kernel pre-pass keeps its (double) registers
    static constexpr int PREPASS_FLOATS_PER_LANE = (O_NET - O_BUFA) / T;
    static constexpr int O_GNET = O_NET + NW * S * T;
    static constexpr int O_ACCB = O_GNET + S * T;
    // bias-gradient rows: one per activation, for any depth the library takes -- except at width 512 (round 6), where 32 rows would be 64 KB
    // of the 160: eight rows, i.e. nets of up to eight 512-wide hidden layers (pinn_create refuses deeper ones)
    static constexpr int ACCB_ROWS = HP >= 512 ? 8 : PINN_MAX_LAYERS;
    static constexpr int O_ACCW1 = O_ACCB + ACCB_ROWS * HP;
    static constexpr int O_SCAL = O_ACCW1 + HP * PINN_XS_LD;
    static constexpr int O_TBAR = O_SCAL;                       // arrival counter of the team-local barrier (PINN_TEAM_FLAGS builds): shares
                                                                // the first slot of `scal`, which is written behind the tile loop only
    static constexpr int O_PREG = O_SCAL + T * 4;
    static constexpr int O_PADJ = O_PREG + PINN_MAX_REGS * T;
    static constexpr int SMEM_FLOATS = O_PADJ + PINN_MAX_REGS * T;
    // fast instantiations may keep W^T of the LH hidden layers in LDS (data-gradient A operand as ds_read_b128):
    static constexpr int O_WT = SMEM_FLOATS;
    static constexpr int WT_LD = HP + 8;
    PINN_HOST_DEVICE static constexpr bool wt_fits(int lh) {
        return lh > 0 && HP <= 64 && !WTG && (SMEM_FLOATS + lh * HP * WT_LD) * 4 <= 160 * 1024;
    }
    PINN_HOST_DEVICE static constexpr int smem_floats(int lh_static) {
        return SMEM_FLOATS + (wt_fits(lh_static) ? lh_static * HP * WT_LD : 0);
    }
    // two-team kernels (VAR 256): each team owns a block [0, O_PREG) of this carve (shape-specialised: no program registers),
    // W^T of the static-depth net sits once behind both blocks
    static constexpr int TEAM_FLOATS = O_PREG;
    // (split-bf16 kernels take their weight fragments from global memory / L2: no W^T block)
    PINN_HOST_DEVICE static constexpr int smem_floats_teams(int lh) { return 2 * TEAM_FLOATS + (SPLIT ? 0 : (lh > 0 ? lh : 0) * HP * WT_LD); }
    PINN_HOST_DEVICE static constexpr bool wt_fits_teams(int lh) { return lh > 0 && HP <= 64 && smem_floats_teams(lh) * 4 <= 160 * 1024; }
    // "slab in LDS" kernels (shape-specialised, affine residual: no program registers): the saved jets take the LDS
    // from O_PREG on -- one value per layer-0 unit, S jets per further activation below the top one (which stays in
    // registers) -- instead of a global slab, and W^T moves from LDS to a per-workgroup global scratch (A.wt)
    PINN_HOST_DEVICE static constexpr int slabl_vec4(int lh) { return (1 + (lh > 1 ? (lh - 1) * S : 0)) * NTW * MT * NTHREADS; }
    PINN_HOST_DEVICE static constexpr int slabl_smem_floats(int lh) { return O_PREG + 4 * slabl_vec4(lh); }
    PINN_HOST_DEVICE static constexpr bool slabl_fits(int lh) {
        return lh >= 1 && HP <= 64 && slabl_smem_floats(lh) * 4 <= 160 * 1024;
    }
    // split-bf16 weight fragments (PinnKArgs::wsp): [layer][direction: 0 forward W, 1 data gradient W^T][K block of 32][16-unit
    // tile j][plane][lane] x 16 bytes -- a wave's A operand of one K block is three coalesced 1-KB loads
    static constexpr int SP_KB = HP / 32;
    PINN_HOST_DEVICE static constexpr size_t wsp_bytes(int lh) { return (size_t)(lh > 0 ? lh : 0) * 2 * SP_KB * NT * 3 * 1024; }
    PINN_HOST_DEVICE static constexpr size_t wsp_frag(int l, int dir, int kb, int j, int p) {      // index in 16-byte units, + lane
        return ((((size_t)(l * 2 + dir) * SP_KB + kb) * NT + j) * 3 + p) * 64;
    }
    // WGX kernels: gz_a, a = 1 .. lh, of one tile (S jets per hidden->hidden layer)
    PINN_HOST_DEVICE static constexpr size_t gz_vec4_per_tile(int lh) { return (size_t)lh * S * NTW * MT * NTHREADS; }
    // one slot of S jets per activation (+ one per skip connection: the skipped activations, later their gradient)
    PINN_HOST_DEVICE static constexpr size_t slab_vec4_per_wg(int lh, int n_skips = 0) {
        return (size_t)(lh + 1 + n_skips) * S * NTW * MT * NTHREADS;
    }
};

// ------------------------------------------------------------------------------------------------------------
// activation and its derivatives from the activation VALUE (tanh: t, sigmoid: s)
// ------------------------------------------------------------------------------------------------------------
// tanh(z) = sign(z) (1 - t)/(1 + t) with t = e^{-2|z|}, sigmoid(z) = 1/(1 + e^{-z}) on v_exp_f32 / v_rcp_f32 (about 1 ulp
// each): absolute error <= ~1.5e-7 over the whole range, saturates cleanly, no branches.
// sin and cos of one argument together (activation 'Sin': value and derivatives come from both): k = nearest integer to
// x * 2/pi, r = x - k * pi/2 by three-constant Cody-Waite steps (exact products for |k| < 2^15, i.e. |x| < 5e4), minimax
// polynomials on [-pi/4, pi/4] (cephes sinf / cosf), quadrant fix-up: ~25 instructions and ~1 ulp, against two to three ocml
// sinf / cosf calls per use (each with its own range reduction and a branch to the large-argument path): the Sin breadth
// workload of bench.py went 0.644 -> see DESIGN.md section 6b.
PINN_DEVICE void pinn_sincos(float x, float& sn, float& cs) {
    const float kf = rintf(x * 0.63661977236758134f);
    float r = fmaf(kf, -1.5703125f, x);
    r = fmaf(kf, -4.837512969970703125e-4f, r);
    r = fmaf(kf, -7.54978995489188216e-8f, r);
    const int k = (int)kf;
    const float r2 = r * r;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f) * r2, r, r);
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f) * r2, r2,
                          fmaf(-0.5f, r2, 1.0f));
    const float a = (k & 1) ? cp : sp, b = (k & 1) ? sp : cp;
    sn = (k & 2) ? -a : a;
    cs = ((k + 1) & 2) ? -b : b;
}
// TIMING ablations of experiment builds (tools/variant.sh -DPINN_ABL=bits; results of such a build are meaningless): 1 the split
// without its arithmetic, 2 no slab stores, 4 no slab loads, 8 no LDS writes of the split planes, 16 no barriers in the tile loop,
// 32 activation derivatives without transcendental functions -- what each piece of the vector phases costs
#ifndef PINN_ABL
#define PINN_ABL 0
#endif
// (bit 8 of the code handed to pinn_act / pinn_jet_fwd: tanh of small arguments by its minimax polynomial -- the kernels that can pay
//  for it, see PTALL in the tile kernel; every other function sees the plain code)
#define PINN_ACT_TANH_POLYBIT 0x100
PINN_DEVICE float pinn_act(float z, PinnAct act_) {
    const bool poly = (act_.c & PINN_ACT_TANH_POLYBIT) != 0;
    const int act = act_.c & 0xff;
    const float par = act_.p;                   // LeakyReLU negative_slope / ELU alpha / Softplus beta
    if (PINN_ABL & 32) return 0.5f * z;
    if (act == PINN_ACT_TANH) {
        // sign(z) (1 - t)/(1 + t) with t = e^{-2|z|} (symmetric, no overflow) for small |z|; 1 - 2t/(1 + t) where the unit saturates
        // (t < 1/2): the subtraction from 1 is then exact and the error of the small second term does not matter, which is what
        // 1 - v^2 downstream needs. Measured on trained models against the fp64 oracle (tools/arbiter.py, DESIGN.md section 6a:
        // gradient error relative to the fp32 reference's own on cfg4 / time on cfg2, cfg4): 1 - 2/(1 + e^{2z}) 2.95x; the first form
        // alone 2.42x (+0 %, +1 %); THIS 1.4x (+0.5 %, +1.9 %); with an odd polynomial below |z| = 0.35 1.82x (+1.1 %, +3.9 %); a
        // minimax polynomial below 0.45 1.23x (+0.6 %, +2.7 %); ocml tanhf 1.29x (+5 %, +8.5 %).
        const float t = pinn_exp2(fabsf(z) * -2.8853900817779268f);
        const float r = pinn_rcp(1.0f + t);
        const float lo = (1.0f - t) * r, hi = 1.0f - (t + t) * r;
#ifndef PINN_TANH_POLY
#define PINN_TANH_POLY 0        // 1: in every kernel (experiment builds); the product asks for it per kernel with PINN_ACT_TANH_POLYBIT
#endif
        if (PINN_TANH_POLY || poly) {
            // |z| < 0.45: z P(z^2), P of degree 4 (Chebyshev fit of tanh(sqrt(w)) / sqrt(w) on [0, 0.2025]; 1.5e-7 relative in fp32
            // evaluation, where the exponential form loses relative accuracy to the cancellation in 1 - t)
            const float w = z * z;
            const float p = z * fmaf(fmaf(fmaf(fmaf(0.017927762120962143f, w, -0.05330030247569084f), w, 0.13328608870506287f), w,
                                          -0.3333321511745453f), w, 1.0f);
            return fabsf(z) < 0.45f ? p : copysignf(t < 0.5f ? hi : lo, z);
        }
        return copysignf(t < 0.5f ? hi : lo, z);
    }
    if (act == PINN_ACT_SIGMOID) return pinn_rcp(1.0f + pinn_exp2(z * -1.4426950408889634f));
    if (act == PINN_ACT_SIN) { float sn, cs; pinn_sincos(z, sn, cs); return sn; }
    if (act == PINN_ACT_SOFTPLUS) return par * z > 20.0f ? z : log1pf(expf(par * z)) / par;    // torch.nn.Softplus (beta, threshold 20)
    if (act == PINN_ACT_SILU) return z / (1.0f + expf(-z));
    if (act == PINN_ACT_GELU) return 0.5f * z * (1.0f + erff(z * 0.70710678118654752f));   // torch.nn.GELU (erf form)
    // round 5: the torch-default forms of the other common `nn` activations (include/pinn.h)
    if (act == PINN_ACT_RELU) return z > 0.0f ? z : 0.0f;
    if (act == PINN_ACT_LEAKYRELU) return z > 0.0f ? z : par * z;
    if (act == PINN_ACT_ELU) return z > 0.0f ? z : par * expm1f(z);
    if (act == PINN_ACT_SELU) return 1.0507009873554805f * (z > 0.0f ? z : 1.6732632423543772f * expm1f(z));
    if (act == PINN_ACT_SOFTSIGN) return z / (1.0f + fabsf(z));
    if (act == PINN_ACT_GELU_TANH) return 0.5f * z * (1.0f + tanhf(0.7978845608028654f * (z + 0.044715f * z * z * z)));
    if (act == PINN_ACT_MISH) return z * tanhf(z > 20.0f ? z : log1pf(expf(z)));
    if (act == PINN_ACT_TANHSHRINK) return z - tanhf(z);
    if (act == PINN_ACT_LOGSIGMOID) return fminf(z, 0.0f) - log1pf(expf(-fabsf(z)));
    return z;                                                     // PINN_ACT_IDENTITY ('f f': no activation in between)
}
// what the reverse half keeps of an activation: its VALUE (tanh, sigmoid, identity: all derivatives follow from it) or the
// pre-activation (sin, softplus, SiLU, GELU: their derivatives are functions of z)
PINN_DEVICE bool pinn_act_keeps_z(int act) { return act == PINN_ACT_SIN || act >= PINN_ACT_SOFTPLUS; }
PINN_DEVICE float pinn_act_saved(float v, float z, int act) { return pinn_act_keeps_z(act) ? z : v; }
PINN_DEVICE float pinn_act_value(float saved, PinnAct act) { return pinn_act_keeps_z(act) ? pinn_act(saved, act) : saved; }
// derivatives 1..4 of the z-keeping smooth activations, from z. With s = sigmoid(z), a = s (1 - s), q = 1 - 2 s (so that
// s' = a, a' = a q, q' = -2 a) and phi = exp(-z^2 / 2) / sqrt(2 pi), Phi' = phi, phi' = -z phi:
//   softplus: s | a | a q | a (q^2 - 2 a)
//   SiLU z s:  s + z a | 2 a + z a q | 3 a q + z a (q^2 - 2 a) | 4 a (q^2 - 2 a) + z a q (q^2 - 8 a)
//   GELU z Phi: Phi + z phi | phi (2 - z^2) | phi z (z^2 - 4) | phi (-z^4 + 7 z^2 - 4)
// derivatives 1..4 of F(z) = tanh(u(z)) from T = tanh(u) and u', .., u'''' (Faa di Bruno); tanh' .. tanh'''' as polynomials in T
PINN_DEVICE void pinn_tanh_chain(float T, float u1, float u2, float u3, float u4, float& t1, float& t2, float& t3, float& t4) {
    const float a1 = 1.0f - T * T, a2 = -2.0f * T * a1, a3 = a1 * (6.0f * T * T - 2.0f), a4 = a1 * T * (16.0f - 24.0f * T * T);
    t1 = a1 * u1;
    t2 = a2 * u1 * u1 + a1 * u2;
    t3 = a3 * u1 * u1 * u1 + 3.0f * a2 * u1 * u2 + a1 * u3;
    t4 = a4 * u1 * u1 * u1 * u1 + 6.0f * a3 * u1 * u1 * u2 + a2 * (4.0f * u1 * u3 + 3.0f * u2 * u2) + a1 * u4;
}
// ... of the round-5 activations (codes >= PINN_ACT_RELU; oracle/jet_f64.py act_derivs states the same formulas in fp64 and
// tests/test_activations.py holds both to torch's nested autograd): piecewise ones as autograd differentiates them (kinks: the
// z > 0 branch decides, every higher derivative of a linear piece is 0), GELU-tanh and Mish as z F(z) with F built on tanh(u(z)):
// (z F)^(n) = z F^(n) + n F^(n-1)
PINN_DEVICE void pinn_act_zderivs_ext(float z, PinnAct act_, float& d1, float& d2, float& d3, float& d4) {
    const int act = act_.c;
    const float par = act_.p;
    d2 = 0.0f; d3 = 0.0f; d4 = 0.0f;
    if (act == PINN_ACT_RELU) { d1 = z > 0.0f ? 1.0f : 0.0f; return; }
    if (act == PINN_ACT_LEAKYRELU) { d1 = z > 0.0f ? 1.0f : par; return; }
    if (act == PINN_ACT_ELU || act == PINN_ACT_SELU) {
        const float sc = act == PINN_ACT_SELU ? 1.0507009873554805f : 1.0f, al = act == PINN_ACT_SELU ? 1.6732632423543772f : par;
        if (z > 0.0f) { d1 = sc; return; }
        const float e = sc * al * expf(z);
        d1 = e; d2 = e; d3 = e; d4 = e;
        return;
    }
    if (act == PINN_ACT_SOFTSIGN) {
        const float a = 1.0f / (1.0f + fabsf(z)), sg = z > 0.0f ? 1.0f : (z < 0.0f ? -1.0f : 0.0f), a2 = a * a;
        d1 = a2; d2 = -2.0f * sg * a2 * a; d3 = 6.0f * a2 * a2; d4 = -24.0f * sg * a2 * a2 * a;
        return;
    }
    if (act == PINN_ACT_TANHSHRINK) {
        float t1, t2, t3, t4;
        pinn_tanh_chain(tanhf(z), 1.0f, 0.0f, 0.0f, 0.0f, t1, t2, t3, t4);
        d1 = 1.0f - t1; d2 = -t2; d3 = -t3; d4 = -t4;
        return;
    }
    // (every case an equality test with its own code: a kernel whose activation codes are masked to the first eight -- ACTC -1,
    //  see act_at -- drops the cases above 7 at compile time)
    if (!(act == PINN_ACT_LOGSIGMOID || act == PINN_ACT_GELU_TANH || act == PINN_ACT_MISH)) { d1 = 1.0f; return; }
    const float sg = 1.0f / (1.0f + expf(-z)), a = sg * (1.0f - sg), q = 1.0f - 2.0f * sg;       // sigmoid and its derivatives a, a q, a (q^2 - 2 a)
    if (act == PINN_ACT_LOGSIGMOID) { d1 = 1.0f - sg; d2 = -a; d3 = -a * q; d4 = -a * (q * q - 2.0f * a); return; }
    float T, u1, u2, u3, u4;
    const bool gelu = act == PINN_ACT_GELU_TANH;
    if (gelu) {
        const float k = 0.7978845608028654f, c = 0.044715f;
        T = tanhf(k * (z + c * z * z * z));
        u1 = k * (1.0f + 3.0f * c * z * z); u2 = 6.0f * k * c * z; u3 = 6.0f * k * c; u4 = 0.0f;
    } else {                                                                                // Mish: u = softplus(z)
        T = tanhf(z > 20.0f ? z : log1pf(expf(z)));
        u1 = sg; u2 = a; u3 = a * q; u4 = a * (q * q - 2.0f * a);
    }
    float t1, t2, t3, t4;
    pinn_tanh_chain(T, u1, u2, u3, u4, t1, t2, t3, t4);
    const float sc = gelu ? 0.5f : 1.0f, f0 = gelu ? 0.5f * (1.0f + T) : T;
    d1 = f0 + sc * z * t1; d2 = sc * (z * t2 + 2.0f * t1); d3 = sc * (z * t3 + 3.0f * t2); d4 = sc * (z * t4 + 4.0f * t3);
}
PINN_DEVICE void pinn_act_zderivs(float z, PinnAct act_, float& d1, float& d2, float& d3, float& d4) {
    const int act = act_.c;
    if (act >= PINN_ACT_RELU) { pinn_act_zderivs_ext(z, act_, d1, d2, d3, d4); return; }
    if (act == PINN_ACT_GELU) {
        const float phi = 0.3989422804014327f * expf(-0.5f * z * z), Phi = 0.5f * (1.0f + erff(z * 0.70710678118654752f));
        const float z2 = z * z;
        d1 = Phi + z * phi; d2 = phi * (2.0f - z2); d3 = phi * z * (z2 - 4.0f); d4 = phi * ((7.0f - z2) * z2 - 4.0f);
        return;
    }
    // (softplus_beta(z) = log(1 + e^{beta z}) / beta: derivative n is beta^{n-1} s^{(n-1)}(beta z))
    const float be = act == PINN_ACT_SOFTPLUS ? act_.p : 1.0f;
    const float sg = 1.0f / (1.0f + expf(-be * z)), a = sg * (1.0f - sg), q = 1.0f - 2.0f * sg;
    const float a3 = a * (q * q - 2.0f * a), a4 = a * q * (q * q - 8.0f * a);       // s''' and s''''
    if (act == PINN_ACT_SOFTPLUS) { d1 = sg; d2 = be * a; d3 = be * be * a * q; d4 = be * be * be * a3; }
    else { d1 = sg + z * a; d2 = 2.0f * a + z * a * q; d3 = 3.0f * a * q + z * a3; d4 = 4.0f * a3 + z * a4; }   // SiLU
}
PINN_DEVICE void pinn_act_d12(float sv, PinnAct act, float& d1, float& d2) {
    if (act == PINN_ACT_TANH) { d1 = 1.0f - sv * sv; d2 = -2.0f * sv * d1; }
    else if (act == PINN_ACT_SIGMOID) { d1 = sv * (1.0f - sv); d2 = d1 * (1.0f - 2.0f * sv); }
    else if (act == PINN_ACT_SIN) { float sn; pinn_sincos(sv, sn, d1); d2 = -sn; }
    else if (act >= PINN_ACT_SOFTPLUS) { float d3, d4; pinn_act_zderivs(sv, act, d1, d2, d3, d4); }
    else { d1 = 1.0f; d2 = 0.0f; }
}
PINN_DEVICE float pinn_act_d3(float sv, float d1, float d2, PinnAct act) {
    if (act == PINN_ACT_TANH) return d1 * (6.0f * sv * sv - 2.0f);
    if (act == PINN_ACT_SIGMOID) {
        const float q = 1.0f - 2.0f * sv;
        return d1 * (q * q - 2.0f * d1);
    }
    if (act == PINN_ACT_SIN) return -d1;
    if (act >= PINN_ACT_SOFTPLUS) { float e1, e2, d3, d4; pinn_act_zderivs(sv, act, e1, e2, d3, d4); return d3; }
    return 0.0f;
}

// fourth derivative from the activation value (reverse sweep of third-order streams)
PINN_DEVICE float pinn_act_d4(float sv, float d1, float d2, PinnAct act) {
    if (act == PINN_ACT_TANH) return d1 * sv * (16.0f - 24.0f * sv * sv);
    if (act == PINN_ACT_SIGMOID) {
        const float q = 1.0f - 2.0f * sv;                        // with a = s(1 - s): s2 = a q, s3 = a (q^2 - 2 a), s4 = a q (q^2 - 8 a)
        return d1 * q * (q * q - 8.0f * d1);
    }
    if (act == PINN_ACT_SIN) return -d2;                          // sin(z) = -(second derivative)
    if (act >= PINN_ACT_SOFTPLUS) { float e1, e2, e3, d4; pinn_act_zderivs(sv, act, e1, e2, e3, d4); return d4; }
    return 0.0f;
}

// fifth derivative (reverse sweep of FOURTH-order streams, round 5). tanh / sigmoid / sin / identity from the activation value like d3 / d4;
// the z-keeping ones from the pre-activation: sigmoid's derivatives s1 = a, s2 = a q, s3 = a (q^2 - 2 a), s4 = a q (q^2 - 8 a),
// s5 = a (q^4 - 22 a q^2 + 16 a^2) give softplus (s4), SiLU (z s5 + 5 s4), LogSigmoid (-s4); GELU z Phi: phi (z^5 - 11 z^3 + 18 z);
// GELU-tanh / Mish / Tanhshrink: Faa di Bruno's fifth term on tanh(u(z)), tanh^(5) = T1 (16 - 120 T^2 + 120 T^4)
PINN_DEVICE float pinn_act_d5(float sv, float d1, float d2, PinnAct act_) {
    const int act = act_.c;
    if (act == PINN_ACT_TANH) { const float t2 = sv * sv; return d1 * (16.0f + t2 * (-120.0f + 120.0f * t2)); }
    if (act == PINN_ACT_SIGMOID) {
        const float q = 1.0f - 2.0f * sv, q2 = q * q;
        return d1 * (q2 * q2 - 22.0f * d1 * q2 + 16.0f * d1 * d1);
    }
    if (act == PINN_ACT_SIN) return d1;                            // cos
    if (act < PINN_ACT_SOFTPLUS) return 0.0f;                      // identity
    const float z = sv;
    if (act == PINN_ACT_RELU || act == PINN_ACT_LEAKYRELU) return 0.0f;
    if (act == PINN_ACT_ELU || act == PINN_ACT_SELU) {
        const float sc = act == PINN_ACT_SELU ? 1.0507009873554805f : 1.0f, al = act == PINN_ACT_SELU ? 1.6732632423543772f : act_.p;
        return z > 0.0f ? 0.0f : sc * al * expf(z);
    }
    if (act == PINN_ACT_SOFTSIGN) { const float a = 1.0f / (1.0f + fabsf(z)), a2 = a * a; return 120.0f * a2 * a2 * a2; }
    if (act == PINN_ACT_GELU) {
        const float phi = 0.3989422804014327f * expf(-0.5f * z * z), z2 = z * z;
        return phi * z * ((z2 - 11.0f) * z2 + 18.0f);
    }
    const float be = act == PINN_ACT_SOFTPLUS ? act_.p : 1.0f;
    const float sg = 1.0f / (1.0f + expf(-be * z)), a = sg * (1.0f - sg), q = 1.0f - 2.0f * sg, q2 = q * q;
    const float s4 = a * q * (q2 - 8.0f * a), s5 = a * (q2 * q2 - 22.0f * a * q2 + 16.0f * a * a);
    if (act == PINN_ACT_SOFTPLUS) return be * be * be * be * s4;
    if (act == PINN_ACT_SILU) return z * s5 + 5.0f * s4;
    if (act == PINN_ACT_LOGSIGMOID) return -s4;
    // tanh(u(z)): u = z (Tanhshrink), k (z + c z^3) (GELU-tanh), softplus(z) (Mish)
    float T, u1, u2, u3, u4, u5;
    if (act == PINN_ACT_TANHSHRINK) { T = tanhf(z); u1 = 1.0f; u2 = u3 = u4 = u5 = 0.0f; }
    else if (act == PINN_ACT_GELU_TANH) {
        const float k = 0.7978845608028654f, c = 0.044715f;
        T = tanhf(k * (z + c * z * z * z));
        u1 = k * (1.0f + 3.0f * c * z * z); u2 = 6.0f * k * c * z; u3 = 6.0f * k * c; u4 = 0.0f; u5 = 0.0f;
    } else {
        T = tanhf(z > 20.0f ? z : log1pf(expf(z)));
        u1 = sg; u2 = a; u3 = a * q; u4 = a * (q2 - 2.0f * a); u5 = s4;
    }
    const float T2 = T * T;
    const float a1 = 1.0f - T2, a2 = -2.0f * T * a1, a3 = a1 * (6.0f * T2 - 2.0f), a4 = a1 * T * (16.0f - 24.0f * T2),
                a5 = a1 * (16.0f + T2 * (-120.0f + 120.0f * T2));
    const float t5 = a5 * u1 * u1 * u1 * u1 * u1 + 10.0f * a4 * u1 * u1 * u1 * u2 + a3 * (15.0f * u1 * u2 * u2 + 10.0f * u1 * u1 * u3)
                     + a2 * (10.0f * u2 * u3 + 5.0f * u1 * u4) + a1 * u5;
    if (act == PINN_ACT_TANHSHRINK) return -t5;
    float t1, t2, t3, t4;
    pinn_tanh_chain(T, u1, u2, u3, u4, t1, t2, t3, t4);
    return (act == PINN_ACT_GELU_TANH ? 0.5f : 1.0f) * (z * t5 + 5.0f * t4);
}

// A differentiation direction is an input column c or a diagonal e_a + e_b / e_a - e_b of two columns (mixed partials by
// polarisation: u_ab = (u_vv - u_aa - u_bb) / 2 with v = e_a + e_b; round 5, mixed THIRD-order partials from third derivatives along
// both diagonals: u_aab = (D3_{a+b} - D3_{a-b} - 2 u_bbb) / 6, u_abb = (D3_{a+b} + D3_{a-b} - 2 u_aaa) / 6).
// Round 6: the WEIGHTED diagonals 2 e_a + e_b / 2 e_a - e_b (PINN_DIR_DOUBLE): the odd part in b of the fourth derivative along
// alpha e_a + e_b is 8 alpha^3 u_aaab + 8 alpha u_abbb, so alpha = 1 and alpha = 2 separate the two:
// u_aaab = (B - 2 A) / 48, u_abbb = (8 A - B) / 48 with A = D4_{a+b} - D4_{a-b}, B = D4_{2a+b} - D4_{2a-b}.
// Code: a | (b + 1) << 4 | minus << 8 | double << 9, b + 1 == 0 for a single column (include/pinn.h PINN_DIR_MINUS, PINN_DIR_DOUBLE).
PINN_DEVICE int pinn_dir_a(int code) { return code & 15; }
PINN_DEVICE int pinn_dir_b(int code) { return ((code >> 4) & 15) - 1; }
PINN_DEVICE float pinn_dir_sb(int code) { return (code & 0x100) ? -1.0f : 1.0f; }        // weight of column b in the direction
// The two extensions of round 6 only exist where the host can send them: weighted diagonals with a FOURTH-order stream (pinn_n4 > 0), a
// third column with a THIRD-order stream (pinn_n3 > 0). Every helper takes that as a compile-time mask X (bit 0: weighted, bit 1: third
// column; pinn_dir_x(N2P) of the instantiation), so that the kernels of every other stream shape compile the decoding away (the third
// column's terms cost the width-256 program kernel of bench.py `gelu256` 2.8 % when they were unconditional).
PINN_HOST_DEVICE constexpr int pinn_dir_x(int n2p) { return (pinn_n4(n2p) > 0 ? 1 : 0) | (pinn_n3(n2p) > 0 ? 2 : 0); }
template <int X = 3> PINN_DEVICE float pinn_dir_wa(int code) { return ((X & 1) && (code & 0x200)) ? 2.0f : 1.0f; }         // weight of column a
// ... and a THIRD column, e_a +- e_b +- e_c (round 6: partials of three different columns, u_abc = [D3_{+,+} - D3_{+,-} - D3_{-,+} + D3_{-,-}] / 24
// over the four sign pairs of b and c): (c + 1) << 10, PINN_DIR_MINUS_C for -e_c
template <int X = 3> PINN_DEVICE int pinn_dir_c(int code) { return (X & 2) ? ((code >> 10) & 15) - 1 : -1; }
PINN_DEVICE float pinn_dir_sc(int code) { return (code & 0x4000) ? -1.0f : 1.0f; }
template <int X = 3> PINN_DEVICE bool pinn_dir_has(int code, int c) { return pinn_dir_a(code) == c || pinn_dir_b(code) == c || pinn_dir_c<X>(code) == c; }
// weight of input column c in the direction: 1 (or 2) for a, +-1 for b and for the third column, 0 otherwise
template <int X = 3> PINN_DEVICE float pinn_dir_coef(int code, int c) {
    return pinn_dir_a(code) == c ? pinn_dir_wa<X>(code) : (pinn_dir_b(code) == c ? pinn_dir_sb(code) : (pinn_dir_c<X>(code) == c ? pinn_dir_sc(code) : 0.0f));
}
// first-layer pre-activation derivative along a direction: weighted sum of the weight columns it contains
template <int X = 3> PINN_DEVICE float pinn_dir_weight(const float* w1row, int code) {
    const int b = pinn_dir_b(code), c = pinn_dir_c<X>(code);
    return pinn_dir_wa<X>(code) * w1row[pinn_dir_a(code)] + (b >= 0 ? pinn_dir_sb(code) * w1row[b] : 0.0f) + (c >= 0 ? pinn_dir_sc(code) * w1row[c] : 0.0f);
}

// Second-order streams. Standard form: stream 1+ND+k is d2/dx_k2 for k < N2. COMB form (N2 == 1): ONE stream
// sum_k c_k d2/dx_k2 over all ND directions with run-time weights c_k (a Laplacian / wave / heat operator needs only
// that combination, which saves N2-1 streams through every GEMM). Both are instances of "second stream j collects
// direction k with weight w": the helpers below give the stream index and weight of direction k.
template <int ND, int N2P, bool COMB>
struct PinnJet {
    static constexpr int N2 = pinn_n2(N2P), N3 = pinn_n3(N2P), N4 = pinn_n4(N2P);
    static constexpr int S = pinn_ns(ND, N2P);
    static_assert(!COMB || N3 == 0, "third-order streams do not combine");
    static_assert(N4 <= N3 && N3 <= N2 && N2 <= ND, "fourth- / third-order directions are the first of the third- / second-order ones");
    static PINN_DEVICE int idx3(int k) { return 1 + ND + N2 + k; }         // third derivative along direction k < N3
    static PINN_DEVICE int idx4(int k) { return 1 + ND + N2 + N3 + k; }    // fourth derivative along direction k < N4
    static PINN_DEVICE bool has2(int k) { return COMB ? true : k < N2; }
    static PINN_DEVICE int idx2(int k) { return COMB ? 1 + ND : 1 + ND + k; }
    static PINN_DEVICE float w(int k, const float* cw) { return COMB ? cw[k] : 1.0f; }
};

// forward jet of one (point, unit): z[S] pre-activations -> h[S] activations
template <int ND, int N2P, bool COMB = false>
PINN_DEVICE void pinn_jet_fwd(const float (&z)[pinn_ns(ND, N2P)], PinnAct act_, float (&h)[pinn_ns(ND, N2P)], const float* cw = nullptr) {
    using J = PinnJet<ND, N2P, COMB>;
    constexpr int N2 = J::N2, N3 = J::N3, N4 = J::N4;
    const float v = pinn_act(z[0], act_);
    const PinnAct act(act_.c & 0xff, act_.p);   // (PINN_ACT_TANH_POLYBIT concerns the value only)
    float d1, d2;
    pinn_act_d12(pinn_act_saved(v, z[0], act), act, d1, d2);
    h[0] = v;
#pragma unroll
    for (int k = 0; k < ND; ++k) h[1 + k] = d1 * z[1 + k];
#pragma unroll
    for (int j = 0; j < N2; ++j) h[1 + ND + j] = d1 * z[1 + ND + j];
#pragma unroll
    for (int k = 0; k < ND; ++k)
        if (J::has2(k)) h[J::idx2(k)] += d2 * J::w(k, cw) * z[1 + k] * z[1 + k];
    if (N3 > 0) {
        // third order along direction k < N3: h''' = s' z''' + 3 s'' z' z'' + s''' z'^3
        const float d3 = pinn_act_d3(pinn_act_saved(v, z[0], act), d1, d2, act);
#pragma unroll
        for (int k = 0; k < N3; ++k) {
            const float z1 = z[1 + k], z2 = z[1 + ND + k];
            h[J::idx3(k)] = d1 * z[J::idx3(k)] + 3.0f * d2 * z1 * z2 + d3 * z1 * z1 * z1;
        }
        if (N4 > 0) {
            // fourth order (round 5): h'''' = s' z'''' + s'' (4 z' z''' + 3 z''^2) + 6 s''' z'^2 z'' + s'''' z'^4
            const float d4 = pinn_act_d4(pinn_act_saved(v, z[0], act), d1, d2, act);
#pragma unroll
            for (int k = 0; k < N4; ++k) {
                const float z1 = z[1 + k], z2 = z[1 + ND + k], z3 = z[J::idx3(k)];
                h[J::idx4(k)] = d1 * z[J::idx4(k)] + d2 * (4.0f * z1 * z3 + 3.0f * z2 * z2) + 6.0f * d3 * z1 * z1 * z2 + d4 * z1 * z1 * z1 * z1;
            }
        }
    }
}

// activations h[S] recomputed from the saved form (v, z_k, z_kk)
template <int ND, int N2P, bool COMB = false>
PINN_DEVICE void pinn_jet_recompute(const float (&sv)[pinn_ns(ND, N2P)], PinnAct act, float (&h)[pinn_ns(ND, N2P)],
                                    const float* cw = nullptr) {
    using J = PinnJet<ND, N2P, COMB>;
    constexpr int N2 = J::N2, N3 = J::N3, N4 = J::N4;
    float d1, d2;
    pinn_act_d12(sv[0], act, d1, d2);
    h[0] = pinn_act_value(sv[0], act);
#pragma unroll
    for (int k = 0; k < ND; ++k) h[1 + k] = d1 * sv[1 + k];
#pragma unroll
    for (int j = 0; j < N2; ++j) h[1 + ND + j] = d1 * sv[1 + ND + j];
#pragma unroll
    for (int k = 0; k < ND; ++k)
        if (J::has2(k)) h[J::idx2(k)] += d2 * J::w(k, cw) * sv[1 + k] * sv[1 + k];
    if (N3 > 0) {
        const float d3 = pinn_act_d3(sv[0], d1, d2, act);
#pragma unroll
        for (int k = 0; k < N3; ++k) {
            const float z1 = sv[1 + k], z2 = sv[1 + ND + k];
            h[J::idx3(k)] = d1 * sv[J::idx3(k)] + 3.0f * d2 * z1 * z2 + d3 * z1 * z1 * z1;
        }
        if (N4 > 0) {
            const float d4 = pinn_act_d4(sv[0], d1, d2, act);
#pragma unroll
            for (int k = 0; k < N4; ++k) {
                const float z1 = sv[1 + k], z2 = sv[1 + ND + k], z3 = sv[J::idx3(k)];
                h[J::idx4(k)] = d1 * sv[J::idx4(k)] + d2 * (4.0f * z1 * z3 + 3.0f * z2 * z2) + 6.0f * d3 * z1 * z1 * z2 + d4 * z1 * z1 * z1 * z1;
            }
        }
    }
}

// reverse jet: gh[S] = dL/dh streams -> gz[S] = dL/dz streams
template <int ND, int N2P, bool COMB = false>
PINN_DEVICE void pinn_jet_bwd(const float (&gh)[pinn_ns(ND, N2P)], const float (&sv)[pinn_ns(ND, N2P)], PinnAct act,
                              float (&gz)[pinn_ns(ND, N2P)], const float* cw = nullptr) {
    using J = PinnJet<ND, N2P, COMB>;
    constexpr int N2 = J::N2, N3 = J::N3, N4 = J::N4;
    const float v = sv[0];
    float d1, d2;
    pinn_act_d12(v, act, d1, d2);
    const float d3 = pinn_act_d3(v, d1, d2, act);
    float acc = d1 * gh[0];
#pragma unroll
    for (int j = 0; j < N2; ++j) {
        gz[1 + ND + j] = d1 * gh[1 + ND + j];
        acc += d2 * sv[1 + ND + j] * gh[1 + ND + j];
    }
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        const float zk = sv[1 + k];
        float gzk = d1 * gh[1 + k];
        acc += d2 * zk * gh[1 + k];
        if (J::has2(k)) {
            const float ghkk = gh[J::idx2(k)] * J::w(k, cw);
            gzk += 2.0f * d2 * zk * ghkk;
            acc += d3 * zk * zk * ghkk;
        }
        gz[1 + k] = gzk;
    }
    if (N3 > 0) {
        // adjoint of h''' = s' z''' + 3 s'' z' z'' + s''' z'^3 (the activation's derivatives depend on z0 as well)
        const float d4 = pinn_act_d4(v, d1, d2, act);
#pragma unroll
        for (int k = 0; k < N3; ++k) {
            const float z1 = sv[1 + k], z2 = sv[1 + ND + k], z3 = sv[J::idx3(k)], g3 = gh[J::idx3(k)];
            gz[J::idx3(k)] = d1 * g3;
            gz[1 + ND + k] += 3.0f * d2 * z1 * g3;
            gz[1 + k] += (3.0f * d3 * z1 * z1 + 3.0f * d2 * z2) * g3;
            acc += (d4 * z1 * z1 * z1 + 3.0f * d3 * z1 * z2 + d2 * z3) * g3;
        }
        if (N4 > 0) {
            // adjoint of h'''' = s' z'''' + s'' (4 z' z''' + 3 z''^2) + 6 s''' z'^2 z'' + s'''' z'^4
            const float d5 = pinn_act_d5(v, d1, d2, act);
#pragma unroll
            for (int k = 0; k < N4; ++k) {
                const float z1 = sv[1 + k], z2 = sv[1 + ND + k], z3 = sv[J::idx3(k)], z4 = sv[J::idx4(k)], g4 = gh[J::idx4(k)];
                gz[J::idx4(k)] = d1 * g4;
                gz[J::idx3(k)] += 4.0f * d2 * z1 * g4;
                gz[1 + ND + k] += (6.0f * d2 * z2 + 6.0f * d3 * z1 * z1) * g4;
                gz[1 + k] += (4.0f * d2 * z3 + 12.0f * d3 * z1 * z2 + 4.0f * d4 * z1 * z1 * z1) * g4;
                acc += (d2 * z4 + d3 * (4.0f * z1 * z3 + 3.0f * z2 * z2) + 6.0f * d4 * z1 * z1 * z2 + d5 * z1 * z1 * z1 * z1) * g4;
            }
        }
    }
    gz[0] = acc;
}

// ------------------------------------------------------------------------------------------------------------
// residual program interpreter (one thread per point; registers live in LDS: reg r of point p at regs[r*T + p])
// ------------------------------------------------------------------------------------------------------------
PINN_DEVICE float pinn_sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

PINN_DEVICE float pinn_prog_forward(const pinn_program_t& pg, float* regs, int T) {
    int last = 0;
    for (int i = 0; i < pg.n_ops; ++i) {
        const unsigned w = pg.code[i];
        const int op = w & 255, dst = (w >> 8) & 255, a = (w >> 16) & 255, b = (w >> 24) & 255;
        const float x = (op == PINN_OP_CONST) ? 0.0f : regs[a * T];
        float y;
        switch (op) {
            case PINN_OP_CONST: y = pg.consts[a]; break;
            case PINN_OP_ADD: y = x + regs[b * T]; break;
            case PINN_OP_SUB: y = x - regs[b * T]; break;
            case PINN_OP_MUL: y = x * regs[b * T]; break;
            case PINN_OP_DIV: y = x / regs[b * T]; break;
            case PINN_OP_NEG: y = -x; break;
            case PINN_OP_SIN: y = sinf(x); break;
            case PINN_OP_COS: y = cosf(x); break;
            case PINN_OP_EXP: y = expf(x); break;
            case PINN_OP_LOG: y = logf(x); break;
            case PINN_OP_TANH: y = tanhf(x); break;
            case PINN_OP_SQRT: y = sqrtf(x); break;
            case PINN_OP_POW: y = powf(x, pg.consts[b]); break;
            case PINN_OP_ABS: y = fabsf(x); break;
            case PINN_OP_SIGMOID: y = pinn_sigmoidf(x); break;
            case PINN_OP_RECIP: y = 1.0f / x; break;
            default: y = x; break;    // COPY
        }
        regs[dst * T] = y;
        last = dst;
    }
    return regs[last * T];
}

// x-only pre-pass of ONE point inside the tile kernel: registers 0..d-1 = the input columns, PINN_OP_STORE writes a
// register to aux row b (read back by the same thread in pinn_point_prefetch). Same arithmetic as pinn_aux_kernel.
// Round 6: in DOUBLE precision -- the columns promoted exactly, the constants unrounded (c64), every operation and elementary
// function in fp64, one rounding to fp32 at the STORE. In fp32 a source term like e pi cos(e pi x) carries a systematic error (the
// rounded pi moves the cosine's argument the same way in every point) that survives the batch sum of gradients that are cancelling
// sums: BASELINE config 4's d(loss)/d(b_L) was 1.1 - 1.4e-5 from the fp64 oracle for that reason alone (include/pinn.h,
// tools/cfg4_bl_probe.py). A handful of operations per point and step.
// `regs` / T: the register file -- LDS of the workgroup (register r of this thread at regs[r * T], T = threads; the host checked
// that a tile's worth of points fits, run_train) or private memory (pinn_aux_kernel: T = 1).
// sin and cos of a double for the pre-pass: Cody-Waite reduction with pi/2 in two parts (fdlibm's medium path: k * pio2_1 is exact for
// |k| < 2^20), fdlibm's kernel polynomials on [-pi/4, pi/4] (1e-16) -- a third of the instructions of ocml's sin / cos, which carry the
// large-argument reduction; beyond |x| = 1e5 the library functions take over. (The fp64 pre-pass cost the one-launch fit chunk of BASELINE
// config 1 -- 100 points, 13.9 us per iteration -- most of 1.3 us per iteration with the library forms: profiles/r06_small_fit_rate.txt.)
PINN_DEVICE void pinn_sincos_f64(double x, double& sn, double& cs) {
    const double kf = rint(x * 6.36619772367581382433e-01);
    double r = fma(-kf, 1.57079632673412561417e+00, x);
    r = fma(-kf, 6.07710050650619224932e-11, r);
    const int k = (int)kf;
    const double z = r * r;
    const double sp = fma(z * r, fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                    2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03),
                                    -1.66666666666666324348e-01), r);
    const double cp = fma(z * z, fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                    -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03),
                                    4.16666666666666019037e-02), fma(-0.5, z, 1.0));
    const double a = (k & 1) ? cp : sp, b = (k & 1) ? sp : cp;
    sn = (k & 2) ? -a : a;
    cs = ((k + 1) & 2) ? -b : b;
}
PINN_DEVICE double pinn_sin_f64(double x) { if (!(fabs(x) < 1e5)) return sin(x); double s, c; pinn_sincos_f64(x, s, c); return s; }
PINN_DEVICE double pinn_cos_f64(double x) { if (!(fabs(x) < 1e5)) return cos(x); double s, c; pinn_sincos_f64(x, s, c); return c; }

// The pre-pass runs ONCE per workgroup (or, in the one-launch fit chunk of the narrow nets, once per iteration) between code that fills
// the instruction cache with something else, and its fp64 library forms (exp, log, tanh, pow, the large-argument sin / cos) are each
// hundreds of instructions: laid out as one switch the dispatch of a seven-operation source term walked a 10 KB region and paid
// ~1 000 cycles per OPERATION -- 7.4 K ticks (3.2 us) per pass, half of what the tile loop of a 16-wide net takes (phase clocks of the
// fit chunk, round 6: profiles/r06_fit_chunk_phases.txt). The operations source terms are made of -- sums, products, sines and cosines of
// moderate arguments -- therefore sit in front, compact; everything else is marked unlikely so that the compiler places it behind them.
PINN_DEVICE double pinn_prepass_rare(int op, double xa, double xb, double cb) {
    switch (op) {
        case PINN_OP_DIV: return xa / xb;
        case PINN_OP_SIN: return sin(xa);           // (|x| >= 1e5: the library's reduction)
        case PINN_OP_COS: return cos(xa);
        case PINN_OP_EXP: return exp(xa);
        case PINN_OP_LOG: return log(xa);
        case PINN_OP_TANH: return tanh(xa);
        case PINN_OP_SQRT: return sqrt(xa);
        case PINN_OP_POW: return pow(xa, cb);
        case PINN_OP_ABS: return fabs(xa);
        case PINN_OP_SIGMOID: return 1.0 / (1.0 + exp(-xa));
        case PINN_OP_RECIP: return 1.0 / xa;
        default: return xa;    // COPY
    }
}
PINN_DEVICE double pinn_prepass_op(int op, double xa, double xb, double cb) {
    if (op == PINN_OP_ADD) return xa + xb;
    if (op == PINN_OP_MUL) return xa * xb;
    if (op == PINN_OP_SUB) return xa - xb;
    if (op == PINN_OP_NEG) return -xa;
    if ((op == PINN_OP_SIN || op == PINN_OP_COS) && __builtin_expect(fabs(xa) < 1e5, 1)) {
        double sn, cs;
        pinn_sincos_f64(xa, sn, cs);
        return op == PINN_OP_SIN ? sn : cs;
    }
    if (op == PINN_OP_COPY) return xa;
    return pinn_prepass_rare(op, xa, xb, cb);
}
PINN_DEVICE void pinn_prepass_point(const pinn_program_t& pg, const PinnPreConsts& c64, const float* x, int d, float* aux, long long n,
                                    long long gi, double* regs, int T) {
    // `regs`: from ONE address space per call site (round 5: a select between the LDS carve and a private array made every register
    // access a flat instruction)
    for (int c = 0; c < d; ++c) regs[c * T] = (double)x[c];
    unsigned w_next = pg.n_ops > 0 ? pg.code[0] : 0u;          // (the next op word is fetched one op ahead: a scalar load per op otherwise)
    // (round 6: the value an operation just produced is what the next one usually reads -- forwarded in a register, so that the chain of
    //  operations is not also a chain of LDS write -> read round trips; the register file itself is written all the same)
    double last = 0.0;
    int last_dst = -1;
    for (int i = 0; i < pg.n_ops; ++i) {
        const unsigned w = w_next;
        if (i + 1 < pg.n_ops) w_next = pg.code[i + 1];
        const int op = w & 255, dst = (w >> 8) & 255, a = (w >> 16) & 255, b = (w >> 24) & 255;
        if (op == PINN_OP_STORE) { aux[(long long)b * n + gi] = (float)((a == last_dst) ? last : regs[a * T]); continue; }
        if (op == PINN_OP_CONST) { last = c64.v[a]; last_dst = dst; regs[dst * T] = last; continue; }
        const bool b_is_reg = op == PINN_OP_ADD || op == PINN_OP_SUB || op == PINN_OP_MUL || op == PINN_OP_DIV;
        const double xa = (a == last_dst) ? last : regs[a * T];
        const double xb = b_is_reg ? ((b == last_dst) ? last : regs[b * T]) : 0.0;
        last = pinn_prepass_op(op, xa, xb, op == PINN_OP_POW ? c64.v[b] : 0.0);
        last_dst = dst;
        regs[dst * T] = last;
    }
}

PINN_DEVICE void pinn_prepass_point_private(const pinn_program_t& pg, const PinnPreConsts& c64, const float* x, int d, float* aux, long long n, long long gi) {
    double priv[PINN_MAX_REGS];                // private (scratch): the program indexes it at run time
    pinn_prepass_point(pg, c64, x, d, aux, n, gi, priv, 1);
}

// reverse sweep: adj[] must be zero on entry for every register; adj[result] is seeded with `seed`.
PINN_DEVICE void pinn_prog_backward(const pinn_program_t& pg, const float* regs, float* adj, int T, float seed = 1.0f) {
    if (pg.n_ops == 0) return;
    adj[((pg.code[pg.n_ops - 1] >> 8) & 255) * T] = seed;
    for (int i = pg.n_ops - 1; i >= 0; --i) {
        const unsigned w = pg.code[i];
        const int op = w & 255, dst = (w >> 8) & 255, a = (w >> 16) & 255, b = (w >> 24) & 255;
        const float g = adj[dst * T];
        adj[dst * T] = 0.0f;                       // registers are single-assignment; clear for the next point
        if (op == PINN_OP_CONST) continue;
        const float x = regs[a * T], y = regs[dst * T];
        switch (op) {
            case PINN_OP_ADD: adj[a * T] += g; adj[b * T] += g; break;
            case PINN_OP_SUB: adj[a * T] += g; adj[b * T] -= g; break;
            case PINN_OP_MUL: { const float xb = regs[b * T]; adj[a * T] += g * xb; adj[b * T] += g * x; } break;
            case PINN_OP_DIV: { const float xb = regs[b * T]; adj[a * T] += g / xb; adj[b * T] -= g * y / xb; } break;
            case PINN_OP_NEG: adj[a * T] -= g; break;
            case PINN_OP_SIN: adj[a * T] += g * cosf(x); break;
            case PINN_OP_COS: adj[a * T] -= g * sinf(x); break;
            case PINN_OP_EXP: adj[a * T] += g * y; break;
            case PINN_OP_LOG: adj[a * T] += g / x; break;
            case PINN_OP_TANH: adj[a * T] += g * (1.0f - y * y); break;
            case PINN_OP_SQRT: adj[a * T] += g * 0.5f / y; break;
            case PINN_OP_POW: { const float e = pg.consts[b]; adj[a * T] += g * e * powf(x, e - 1.0f); } break;
            case PINN_OP_ABS: adj[a * T] += (x >= 0.0f ? g : -g); break;
            case PINN_OP_SIGMOID: adj[a * T] += g * y * (1.0f - y); break;
            case PINN_OP_RECIP: adj[a * T] -= g * y * y; break;
            default: adj[a * T] += g; break;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// point stage: ansatz forward, residual / upstream gradient, ansatz reverse.  One thread per point of the tile.
// Formulas: oracle/jet_f64.py (ansatz_forward / ansatz_backward), i.e. model_torch.py:107-128 + product rule.
// ------------------------------------------------------------------------------------------------------------
template <int ND, int N2P>
struct PinnPointOut {
    float gnet[pinn_ns(ND, N2P)];
    float loss, g_ls;
    float g_ic;        // d(loss)/du of the point = its share of d(loss)/d(constant initial value)
};

// per-point values that come from global memory (pre-pass rows, IC streams): fetched at the START of the tile by the
// point's thread, so that their latency is not paid inside the serial point stage
// Shape facts the point stage and the first layer branch on. SPEC 0 reads all of them from the kernel arguments (any
// problem). SPEC 1..3 fix them at compile time for the commonest training shapes -- always: direction k = input column
// k, training step with an affine residual whose coefficients are constants (source term constant or one pre-pass row):
//   1  Dirichlet box: hard boundary binding over all ND = d inputs, no initial condition           (Poisson, Helmholtz)
//   2  evolution in a box: boundary binding over the first ND-1 inputs, initial condition in the last (heat, wave)
//   3  ODE family: one differentiated input (t), initial condition, no boundary binding, d - 1 parameters
// The launcher checks the arguments against exactly this list before it picks a SPEC instantiation (pinn_spec_of).
// What it buys: dozens of loop-invariant conditions (j < nsp, c < d, "does direction k contain column c", ...) no longer
// live as 64-bit lane masks in SGPRs -- the general form keeps so many of them that hipcc spills SGPRs to VGPR lanes
// (several hundred v_readlane per tile) and the kernel needs ~95 registers more -- and the serial one-thread-per-point
// stage shrinks to the arithmetic it needs (cfg2 kernel: 0.237 -> 0.207 ms).
// Round 6: SPEC | 4 = the same shapes with a residual PROGRAM instead of the affine form (non-affine equations, trainable V(...) scalars in
// the equation): the ansatz facts stay compile-time, the residual kind and the program come from the arguments -- so that the two-team
// kernel of the Poisson-box shape serves them too (VAR 2048; they ran on the general one-wave-per-SIMD kernel before: 46 % MFMA-busy).
template <int SPEC_, int ND>
struct PinnShape {
    static constexpr int SPEC = SPEC_ & 3;
    static constexpr bool PROGRAM = (SPEC_ & 4) != 0;
    static constexpr bool FIXED = SPEC != 0;
    static PINN_DEVICE int d(const PinnKArgs& A) { return (SPEC == 1 || SPEC == 2) ? ND : A.d; }
    static PINN_DEVICE int nsp(const PinnKArgs& A) { return SPEC == 1 ? ND : SPEC == 2 ? ND - 1 : SPEC == 3 ? 0 : A.nsp; }
    static PINN_DEVICE int ndims(const PinnKArgs& A) { return (SPEC == 1 || SPEC == 2) ? ND : SPEC == 3 ? 1 : A.ndims; }
    static PINN_DEVICE bool has_bc(const PinnKArgs& A) { return (SPEC == 1 || SPEC == 2) ? true : SPEC == 3 ? false : A.has_bc != 0; }
    static PINN_DEVICE bool has_ic(const PinnKArgs& A) { return SPEC == 1 ? false : FIXED ? true : A.has_ic != 0; }
    static PINN_DEVICE int dir(const PinnKArgs& A, int k) { return FIXED ? k : A.dir_cols[k]; }
    static PINN_DEVICE int mode(const PinnKArgs& A) { return FIXED ? (int)PINN_MODE_STEP : A.mode; }
    static PINN_DEVICE int res_kind(const PinnKArgs& A) { return FIXED ? (PROGRAM ? (int)PINN_RES_PROGRAM : (int)PINN_RES_AFFINE) : A.res_kind; }
    static PINN_DEVICE int s_user(const PinnKArgs& A) { return FIXED ? 99 : A.s_user; }
    static PINN_DEVICE int coef_row(const PinnKArgs& A, int s) { return FIXED ? -1 : A.coef_row[s]; }
};

template <int ND, int N2P>
struct PinnPointPre {
    float src;                       // affine source term F
    float cs[pinn_ns(ND, N2P)];      // affine coefficients C_s
    float ic[pinn_ns(ND, N2P)];      // IC streams
};

template <int ND, int N2P, int SPEC = 0>
PINN_DEVICE void pinn_point_prefetch(const PinnKArgs& A, const float* params_, long long gidx, bool valid, float* pregs, int T,
                                     PinnPointPre<ND, N2P>& pre, const float* aux_) {
    // aux_: the rows of the x-only pre-pass -- A.aux, or the LDS copy of the one-CU fit chunk (pinn_fit_kernel.h)
    constexpr int S = pinn_ns(ND, N2P);
    using SH = PinnShape<SPEC, ND>;
    const long long gi = valid ? gidx : 0;
    pre.src = A.src_const;
#pragma unroll
    for (int s = 0; s < S; ++s) { pre.cs[s] = 0.0f; pre.ic[s] = 0.0f; }
    if (SH::mode(A) == PINN_MODE_STEP) {
        if (SH::res_kind(A) == PINN_RES_AFFINE) {
            if (A.src_row >= 0) pre.src = aux_[(long long)A.src_row * A.n_points + gi];
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (s < SH::s_user(A))
                    pre.cs[s] = (SH::coef_row(A, s) >= 0) ? aux_[(long long)SH::coef_row(A, s) * A.n_points + gi] : A.coef[s];
        } else {
            for (int m = 0; m < A.n_aux; ++m) pregs[(S + SH::d(A) + m) * T] = aux_[(long long)m * A.n_points + gi];
        }
    }
    if (SH::has_ic(A)) {
        if (A.ic_streams) {
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (s < SH::s_user(A) && valid) pre.ic[s] = A.ic_streams[(long long)s * A.n_points + gidx];
        } else if (A.ic_rows) {
            // callable IC lowered into the x-only pre-pass (value + derivative streams as aux rows / constants)
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (s < SH::s_user(A)) pre.ic[s] = (A.ic_row[s] >= 0) ? aux_[(long long)A.ic_row[s] * A.n_points + gi] : A.ic_cst[s];
        } else {
            pre.ic[0] = (!SH::FIXED && A.ic_var1 > 0) ? params_[A.off_extra + A.ic_var1 - 1] : A.ic_const;
        }
    }
}

template <int ND, int N2P, bool WITH_PROGRAMS = true, bool COMB = false, int SPEC = 0>
PINN_DEVICE void pinn_point_stage(const PinnKArgs& A, const float* params_, const float (&net)[pinn_ns(ND, N2P)], const float* x /*[d]*/,
                                  long long gidx, bool valid, float* pregs, float* padj, int T,
                                  const PinnPointPre<ND, N2P>& pre, PinnPointOut<ND, N2P>& out, float es_in = -1.0f) {
    // es_in >= 0: exp(-log_scale), computed ONCE per launch by the caller (PINN_GATE_FAST: the gate's sigmoid then also takes the hardware
    // exp / rcp the activations use -- the serial point stage of the shape-specialised kernels is a chain of dependent instructions that
    // three of a team's four waves wait for)
    constexpr int S = pinn_ns(ND, N2P), N2 = pinn_n2(N2P), N3 = pinn_n3(N2P), N4 = pinn_n4(N2P);
    using J = PinnJet<ND, N2P, COMB>;
    using SH = PinnShape<SPEC, ND>;
    constexpr int NIN = SH::FIXED ? (ND > 0 ? ND : 1) : PINN_MAX_INPUTS;   // input columns the box factors may range over
    constexpr int DX = pinn_dir_x(N2P);                                    // which direction-code extensions this stream shape can meet
    const float* cw = A.comb_w;
    // ---- BC factor P and its direction derivatives --------------------------------------------------------
    float P = 1.0f, Pk[ND > 0 ? ND : 1], Pkk[ND > 0 ? ND : 1], Pkkk[N3 > 0 ? N3 : 1], P4[N4 > 0 ? N4 : 1];
#pragma unroll
    for (int k = 0; k < ND; ++k) { Pk[k] = 0.0f; Pkk[k] = 0.0f; }
#pragma unroll
    for (int k = 0; k < (N3 > 0 ? N3 : 1); ++k) Pkkk[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < (N4 > 0 ? N4 : 1); ++k) P4[k] = 0.0f;
    if (SH::has_bc(A)) {
        float p[NIN], p1[NIN], p2[NIN];
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            p[j] = 1.0f; p1[j] = 0.0f; p2[j] = 0.0f;
            if (j < SH::nsp(A)) {
                const float lo = A.lo[j], hi = A.hi[j], iw = A.inv_w[j], xj = x[j];
                p[j] = ((xj - lo) * iw) * ((hi - xj) * iw);
                p1[j] = (lo + hi - 2.0f * xj) * (iw * iw);
                p2[j] = -2.0f * (iw * iw);
                P *= p[j];
            }
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            // directional derivatives of P = prod_j p_j along v = wa e_a + w e_b (w = +-1, wa = 1 or 2): with A(t) = p_a(x_a + wa t),
            // B(t) = p_b(x_b + w t) and R = prod_{j != a, b} p_j:   first = (A1 B + A B1) R,   second = (A2 B + 2 A1 B1 + A B2) R,
            // third = 3 (A2 B1 + A1 B2) R  (every factor is quadratic in its column: A3 = B3 = 0; along a single column: 0);
            // A1 = wa p1_a, A2 = wa^2 p2_a, B1 = w p1_b, B2 = p2_b (columns outside the spatial block contribute nothing)
            const int ca = pinn_dir_a(SH::dir(A, k)), cb = pinn_dir_b(SH::dir(A, k));
            const float wb = SH::FIXED ? 1.0f : pinn_dir_sb(SH::dir(A, k));
            const float wa = SH::FIXED ? 1.0f : pinn_dir_wa<DX>(SH::dir(A, k));
            float first = 0.0f, second = 0.0f, cross = 2.0f * wa * wb, third = 0.0f;
            bool both = true;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int c = t == 0 ? ca : cb;
                if (c >= 0 && c < SH::nsp(A)) {
                    float rest = 1.0f, q1 = 0.0f, q2 = 0.0f;
#pragma unroll
                    for (int j = 0; j < NIN; ++j) {
                        if (j == c) { q1 = p1[j]; q2 = p2[j]; }
                        else rest *= p[j];
                    }
                    first += (t == 0 ? wa : wb) * q1 * rest;
                    second += (t == 0 ? wa * wa : 1.0f) * q2 * rest;
                } else {
                    both = false;
                }
            }
            if (both) {
#pragma unroll
                for (int j = 0; j < NIN; ++j) cross *= (j == ca || j == cb) ? p1[j] : p[j];
                second += cross;
                if (N3 > 0 && k < N3) {
                    float rab = 3.0f, a1 = 0.0f, a2 = 0.0f, b1 = 0.0f, b2 = 0.0f;
#pragma unroll
                    for (int j = 0; j < NIN; ++j) {
                        if (j == ca) { a1 = p1[j]; a2 = p2[j]; }
                        else if (j == cb) { b1 = p1[j]; b2 = p2[j]; }
                        else rab *= p[j];
                    }
                    third = rab * (wa * wa * a2 * wb * b1 + wa * a1 * b2);
                    if (N4 > 0 && k < N4) P4[k < N4 ? k : 0] = 2.0f * rab * wa * wa * a2 * b2;      // fourth: 6 A2 B2 R (rab carries the 3)
                }
            }
            if (!SH::FIXED && (DX & 2)) {
                // a THIRD spatial column in the direction, C(t) = p_c(x_c + wc t), wc = +-1 (round 6). On top of the pair's terms:
                //   first  += C1 (A B) R
                //   second += C2 (A B) R + 2 C1 (A1 B + A B1) R
                //   third  += 3 C1 (A2 B + A B2 + 2 A1 B1) R + 3 C2 (A1 B + A B1) R           (A3 = B3 = C3 = 0)
                // with A, B standing for 1 where that column is not a spatial one; fourth order along such a direction is not asked for
                const int cc = pinn_dir_c<DX>(SH::dir(A, k));
                if (cc >= 0 && cc < SH::nsp(A)) {
                    const float wc = pinn_dir_sc(SH::dir(A, k));
                    const bool a_sp = ca >= 0 && ca < SH::nsp(A), b_sp = cb >= 0 && cb < SH::nsp(A);
                    float rest = 1.0f, c0 = 1.0f, c1 = 0.0f, c2 = 0.0f, a0 = 1.0f, a1 = 0.0f, a2 = 0.0f, b0 = 1.0f, b1 = 0.0f, b2 = 0.0f;
#pragma unroll
                    for (int j = 0; j < NIN; ++j) {
                        if (j == cc) { c0 = p[j]; c1 = wc * p1[j]; c2 = p2[j]; }
                        else if (j == ca && a_sp) { a0 = p[j]; a1 = wa * p1[j]; a2 = wa * wa * p2[j]; }
                        else if (j == cb && b_sp) { b0 = p[j]; b1 = wb * p1[j]; b2 = p2[j]; }
                        else rest *= p[j];
                    }
                    const float ab0 = a0 * b0, ab1 = a1 * b0 + a0 * b1, ab2 = a2 * b0 + a0 * b2 + 2.0f * a1 * b1;
                    // (the pair's own terms above were taken with the third column's factor inside `rest`: they carry C already)
                    first += c1 * ab0 * rest;
                    second += (c2 * ab0 + 2.0f * c1 * ab1) * rest;
                    third += 3.0f * (c1 * ab2 + c2 * ab1) * rest;
                    (void)c0;
                }
            }
            Pk[k] = first;
            Pkk[k] = second;
            if (N3 > 0 && k < N3) Pkkk[k < N3 ? k : 0] = third;
        }
    }
    // ---- Q = net * P + bc ---------------------------------------------------------------------------------
    float Q[S];
#pragma unroll
    for (int s = 0; s < S; ++s) Q[s] = net[s];
    if (SH::has_bc(A)) {
        Q[0] = net[0] * P + A.bc_value;
#pragma unroll
        for (int k = 0; k < ND; ++k) Q[1 + k] = net[1 + k] * P + net[0] * Pk[k];
#pragma unroll
        for (int j = 0; j < N2; ++j) Q[1 + ND + j] = net[1 + ND + j] * P;
#pragma unroll
        for (int k = 0; k < ND; ++k)
            if (J::has2(k)) Q[J::idx2(k)] += J::w(k, cw) * (2.0f * net[1 + k] * Pk[k] + net[0] * Pkk[k]);
        // third order (single-column directions: every factor of P is quadratic in its column, so P''' = 0):
        // Q''' = net''' P + 3 net'' P' + 3 net' P''
#pragma unroll
        for (int k = 0; k < N3; ++k)
            Q[J::idx3(k)] = net[J::idx3(k)] * P + 3.0f * (net[1 + ND + k] * Pk[k] + net[1 + k] * Pkk[k]) + net[0] * Pkkk[k];
        // fourth order: Q'''' = net'''' P + 4 net''' P' + 6 net'' P'' + 4 net' P''' + net P''''
#pragma unroll
        for (int k = 0; k < N4; ++k)
            Q[J::idx4(k)] = net[J::idx4(k)] * P + 4.0f * (net[J::idx3(k)] * Pk[k] + net[1 + k] * Pkkk[k]) + 6.0f * net[1 + ND + k] * Pkk[k] + net[0] * P4[k];
    }
    // ---- IC gate G = sigmoid(tau) - 1/2, tau = (t - t0) exp(-log_scale) -----------------------------------------
    float u[S];
#pragma unroll
    for (int s = 0; s < S; ++s) u[s] = Q[s];
    float G = 1.0f, Gk[ND > 0 ? ND : 1], Gkk[ND > 0 ? ND : 1], dG = 0.0f, dGk[ND > 0 ? ND : 1], dGkk[ND > 0 ? ND : 1];
    float Gkkk[N3 > 0 ? N3 : 1], dGkkk[N3 > 0 ? N3 : 1], G4[N4 > 0 ? N4 : 1], dG4[N4 > 0 ? N4 : 1];
#pragma unroll
    for (int k = 0; k < ND; ++k) { Gk[k] = 0.0f; Gkk[k] = 0.0f; dGk[k] = 0.0f; dGkk[k] = 0.0f; }
#pragma unroll
    for (int k = 0; k < (N3 > 0 ? N3 : 1); ++k) { Gkkk[k] = 0.0f; dGkkk[k] = 0.0f; }
#pragma unroll
    for (int k = 0; k < (N4 > 0 ? N4 : 1); ++k) { G4[k] = 0.0f; dG4[k] = 0.0f; }
    if (SH::has_ic(A)) {
        const int tcol = SH::ndims(A) - 1;
        const float es = es_in >= 0.0f ? es_in : expf(-params_[A.off_ls]);
        const float tau = (x[tcol] - A.t0) * es;
        const float sg = es_in >= 0.0f ? pinn_rcp(1.0f + pinn_exp2(tau * -1.4426950408889634f)) : pinn_sigmoidf(tau);
        float d1, d2;
        pinn_act_d12(sg, PINN_ACT_SIGMOID, d1, d2);
        const float d3 = pinn_act_d3(sg, d1, d2, PINN_ACT_SIGMOID);
        G = sg - 0.5f;
        dG = -tau * d1;
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            if (pinn_dir_has<DX>(SH::dir(A, k), tcol)) {
                // (wt: weight of the time column in the direction: -1 as the second column of a minus diagonal, 2 as the first column of a
                //  weighted one; the derivative of order n carries wt^n)
                const float wt = SH::FIXED ? 1.0f : pinn_dir_coef<DX>(SH::dir(A, k), tcol);
                const float wt2 = wt * wt;
                Gk[k] = wt * d1 * es; Gkk[k] = wt2 * d2 * es * es;
                dGk[k] = wt * es * (-tau * d2 - d1);
                dGkk[k] = wt2 * es * es * (-tau * d3 - 2.0f * d2);
                if (k < N3) {
                    // G''' = s'''(tau) es^3 and its derivative with respect to log_scale (d tau / ds = -tau, d es / ds = -es)
                    const float d4 = pinn_act_d4(sg, d1, d2, PINN_ACT_SIGMOID);
                    Gkkk[k] = wt2 * wt * d3 * es * es * es;
                    dGkkk[k] = wt2 * wt * es * es * es * (-tau * d4 - 3.0f * d3);
                    if (N4 > 0 && k < N4) {
                        // G'''' = s''''(tau) es^4 wt^4; d / d log_scale: -tau s^(5) es^4 - 4 s'''' es^4
                        const float d5 = pinn_act_d5(sg, d1, d2, PINN_ACT_SIGMOID), es4 = es * es * es * es * wt2 * wt2;
                        G4[k < N4 ? k : 0] = d4 * es4;
                        dG4[k < N4 ? k : 0] = es4 * (-tau * d5 - 4.0f * d4);
                    }
                }
            }
        }
        u[0] = G * Q[0];
#pragma unroll
        for (int k = 0; k < ND; ++k) u[1 + k] = Gk[k] * Q[0] + G * Q[1 + k];
#pragma unroll
        for (int j = 0; j < N2; ++j) u[1 + ND + j] = G * Q[1 + ND + j];
#pragma unroll
        for (int k = 0; k < ND; ++k)
            if (J::has2(k)) u[J::idx2(k)] += J::w(k, cw) * (Gkk[k] * Q[0] + 2.0f * Gk[k] * Q[1 + k]);
        // u''' = G Q''' + 3 G' Q'' + 3 G'' Q' + G''' Q
#pragma unroll
        for (int k = 0; k < N3; ++k)
            u[J::idx3(k)] = G * Q[J::idx3(k)] + 3.0f * (Gk[k] * Q[1 + ND + k] + Gkk[k] * Q[1 + k]) + Gkkk[k] * Q[0];
        // u'''' = G Q'''' + 4 G' Q''' + 6 G'' Q'' + 4 G''' Q' + G'''' Q
#pragma unroll
        for (int k = 0; k < N4; ++k)
            u[J::idx4(k)] = G * Q[J::idx4(k)] + 4.0f * (Gk[k] * Q[J::idx3(k)] + Gkkk[k] * Q[1 + k]) + 6.0f * Gkk[k] * Q[1 + ND + k] + G4[k] * Q[0];
#pragma unroll
        for (int s = 0; s < S; ++s) u[s] += pre.ic[s];
    }
    // ---- output / residual / upstream gradient -----------------------------------------------------------------
    float gu[S];
#pragma unroll
    for (int s = 0; s < S; ++s) gu[s] = 0.0f;
    out.loss = 0.0f;
    if (SH::mode(A) == PINN_MODE_FORWARD) {
        if (valid) {
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (s < SH::s_user(A)) A.out_streams[(long long)s * A.n_points + gidx] = u[s];
        }
#pragma unroll
        for (int s = 0; s < S; ++s) out.gnet[s] = 0.0f;
        out.g_ls = 0.0f;
        out.g_ic = 0.0f;
        return;
    } else if (SH::mode(A) == PINN_MODE_STEP && SH::res_kind(A) == PINN_RES_AFFINE) {
        // r = sum_s C_s u_s + F, coefficients constant or per-point rows of the x-only pre-pass
        float r = pre.src;
#pragma unroll
        for (int s = 0; s < S; ++s) r = fmaf(pre.cs[s], u[s], r);
        const float w = valid ? 2.0f * r * A.inv_n : 0.0f;
#pragma unroll
        for (int s = 0; s < S; ++s) gu[s] = w * pre.cs[s];
        out.loss = valid ? r * r * A.inv_n : 0.0f;
    } else if (WITH_PROGRAMS && SH::mode(A) == PINN_MODE_STEP) {
        // registers: S streams (the INSTANTIATION's S), d input columns, n_aux pre-pass rows (already staged by
        // pinn_point_prefetch), then temporaries
#pragma unroll
        for (int s = 0; s < S; ++s) pregs[s * T] = u[s];
        for (int c = 0; c < SH::d(A); ++c) pregs[(S + c) * T] = x[c];
        // trainable V(...) scalars: registers behind the aux rows; the adjoints of these registers are never cleared,
        // so they add up d(loss)/dV over all points this thread sees (summed over the tile's threads at the end of the kernel)
        const int vbase = S + SH::d(A) + A.n_aux;
        for (int k = 0; k < A.n_vars; ++k) pregs[(vbase + k) * T] = params_[A.off_extra + k];
        const float r = pinn_prog_forward(A.prog, pregs, T);
        const float w = valid ? 2.0f * r * A.inv_n : 0.0f;
        pinn_prog_backward(A.prog, pregs, padj, T, w);          // seeded with d(loss)/dr: adjoints come out scaled
#pragma unroll
        for (int s = 0; s < S; ++s) { gu[s] = padj[s * T]; padj[s * T] = 0.0f; }
        for (int c = 0; c < SH::d(A) + A.n_aux; ++c) padj[(S + c) * T] = 0.0f;
        out.loss = valid ? r * r * A.inv_n : 0.0f;
    } else {
        if (valid) {
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (s < SH::s_user(A)) gu[s] = A.gin[(long long)s * A.n_points + gidx];
        }
    }
    // ---- reverse: u -> Q (gate) ---------------------------------------------------------------------------------
    float gQ[S];
#pragma unroll
    for (int s = 0; s < S; ++s) gQ[s] = gu[s];
    float g_ls = 0.0f;
    if (SH::has_ic(A)) {
        float gG = gu[0] * Q[0];
        gQ[0] = gu[0] * G;
#pragma unroll
        for (int j = 0; j < N2; ++j) {
            gG += gu[1 + ND + j] * Q[1 + ND + j];
            gQ[1 + ND + j] = gu[1 + ND + j] * G;
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            gG += gu[1 + k] * Q[1 + k];
            float gGk = gu[1 + k] * Q[0];
            gQ[0] += gu[1 + k] * Gk[k];
            gQ[1 + k] = gu[1 + k] * G;
            if (J::has2(k)) {
                const float gkk = gu[J::idx2(k)] * J::w(k, cw);
                gGk += 2.0f * gkk * Q[1 + k];
                g_ls += gkk * Q[0] * dGkk[k];
                gQ[0] += gkk * Gkk[k];
                gQ[1 + k] += 2.0f * gkk * Gk[k];
            }
            g_ls += gGk * dGk[k];
        }
#pragma unroll
        for (int k = 0; k < N3; ++k) {
            const float g3 = gu[J::idx3(k)];
            gQ[J::idx3(k)] = g3 * G;
            gQ[1 + ND + k] += 3.0f * g3 * Gk[k];
            gQ[1 + k] += 3.0f * g3 * Gkk[k];
            gQ[0] += g3 * Gkkk[k];
            gG += g3 * Q[J::idx3(k)];
            g_ls += g3 * (3.0f * (Q[1 + ND + k] * dGk[k] + Q[1 + k] * dGkk[k]) + Q[0] * dGkkk[k]);
        }
#pragma unroll
        for (int k = 0; k < N4; ++k) {
            const float g4 = gu[J::idx4(k)];
            gQ[J::idx4(k)] = g4 * G;
            gQ[J::idx3(k)] += 4.0f * g4 * Gk[k];
            gQ[1 + ND + k] += 6.0f * g4 * Gkk[k];
            gQ[1 + k] += 4.0f * g4 * Gkkk[k];
            gQ[0] += g4 * G4[k];
            gG += g4 * Q[J::idx4(k)];
            g_ls += g4 * (4.0f * (Q[J::idx3(k)] * dGk[k] + Q[1 + k] * dGkkk[k]) + 6.0f * Q[1 + ND + k] * dGkk[k] + Q[0] * dG4[k]);
        }
        g_ls += gG * dG;
    }
    // ---- reverse: Q -> net (BC factor) -----------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < S; ++s) out.gnet[s] = gQ[s];
    if (SH::has_bc(A)) {
        float g0 = gQ[0] * P;
#pragma unroll
        for (int j = 0; j < N2; ++j) out.gnet[1 + ND + j] = gQ[1 + ND + j] * P;
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            g0 += gQ[1 + k] * Pk[k];
            float g1 = gQ[1 + k] * P;
            if (J::has2(k)) {
                const float gkk = gQ[J::idx2(k)] * J::w(k, cw);
                g0 += gkk * Pkk[k];
                g1 += 2.0f * gkk * Pk[k];
            }
            out.gnet[1 + k] = g1;
        }
#pragma unroll
        for (int k = 0; k < N3; ++k) {
            const float g3 = gQ[J::idx3(k)];
            out.gnet[J::idx3(k)] = g3 * P;
            out.gnet[1 + ND + k] += 3.0f * g3 * Pk[k];
            out.gnet[1 + k] += 3.0f * g3 * Pkk[k];
            g0 += g3 * Pkkk[k];             // (third derivative of the box factor along a diagonal; 0 along a single column)
        }
#pragma unroll
        for (int k = 0; k < N4; ++k) {
            const float g4 = gQ[J::idx4(k)];
            out.gnet[J::idx4(k)] = g4 * P;
            out.gnet[J::idx3(k)] += 4.0f * g4 * Pk[k];
            out.gnet[1 + ND + k] += 6.0f * g4 * Pkk[k];
            out.gnet[1 + k] += 4.0f * g4 * Pkkk[k];
            g0 += g4 * P4[k];
        }
        out.gnet[0] = g0;
    }
    out.g_ls = g_ls;
    out.g_ic = gu[0];
}

// ------------------------------------------------------------------------------------------------------------
// the tile kernel
//
// Lane map (all MFMA accumulators, "swapped" form D = W . H^T so that one lane owns 4 CONSECUTIVE units of ONE
// point and every LDS activation write is a ds_write_b128):
//     lr = lane & 15  -> point  pt = 16*mt + lr            lq = lane >> 4 -> unit quad
//     wave w, unit tile j  -> units  n = (w*NTW + j)*16 + 4*lq + r,  r = accumulator component 0..3
// LHC / ACTC >= 0 fix the number of hidden->hidden layers / the activation at compile time (fast instantiations
// for the BASELINE configs); -1 keeps them run-time (generic instantiations).
// ------------------------------------------------------------------------------------------------------------
#if defined(PINN_PROFILE_PHASES) && !defined(PINN_EMU)
#define PH_DECL long long ph_acc[16] = {0}; long long ph_last = __builtin_readcyclecounter();
#define PH(i) { PINN_SCHED_BARRIER(); const long long ph_now = __builtin_readcyclecounter(); ph_acc[i] += ph_now - ph_last; ph_last = ph_now; PINN_SCHED_BARRIER(); }
#define PH_FLUSH if (A.prof && lane == 0) { for (int i = 0; i < 16; ++i) A.prof[((size_t)vbid * NW + wave) * 16 + i] = ph_acc[i]; }      /* (a row per TEAM: vbid) */
#else
#define PH_DECL
#define PH(i)
#define PH_FLUSH
#endif

PINN_DEVICE f32x4 pinn_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// the per-tile slabs of the WGX kernels are written once and read once, by another launch: streaming (non-temporal) accesses
// (same-box A/B, MI355X: at width 256 -- 5.9 GB of slab per 131 072 points -- non-temporal stores and loads take 4.8 % off
//  the two kernels, 9.07 -> 8.64 ms; at width 128 they change nothing, so PINN_SLAB_NT_MIN_HP picks the widths)
#ifndef PINN_SLAB_NT_MIN_HP
#define PINN_SLAB_NT_MIN_HP 256
#endif
template <bool NT> PINN_DEVICE void pinn_st4_stream(f32x4* p, f32x4 v) {
#ifndef PINN_EMU
    if (NT) { __builtin_nontemporal_store(v, p); return; }
#endif
    *p = v;
}
template <bool NT> PINN_DEVICE f32x4 pinn_ld4_stream(const f32x4* p) {
#ifndef PINN_EMU
    if (NT) return __builtin_nontemporal_load(p);
#endif
    return *p;
}
PINN_DEVICE void pinn_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// ---- split-bf16 operands -----------------------------------------------------------------------------------------------
// x = hi + mid + lo EXACTLY, each a bf16: hi = x truncated to its upper 16 bits, mid = (x - hi) truncated, lo = x - hi - mid
// (the subtractions are exact: 24 = 8 + 8 + 8 mantissa bits). Four VALU operations per value, then one v_perm_b32 per PAIR of
// values and plane packs the upper halves (the pack is the truncation of mid and lo).
PINN_DEVICE unsigned pinn_fbits(float x) { return __builtin_bit_cast(unsigned, x); }
PINN_DEVICE float pinn_bitsf(unsigned x) { return __builtin_bit_cast(float, x); }
// hi and mid are ROUNDED to nearest (+0x8000 on the bit pattern before the mask), lo takes the rest exactly. Truncation would be
// two instructions per value cheaper and just as exact a split, but then every part of a value carries the value's sign and
// the matrix pipe's internal accumulation (which does not round to nearest) drifts one way: on TRAINED models, arbitrated in
// fp64, the gradient error came out at 2.18x the fp32 reference's own (cfg4) with truncation, 1.34x with mid alone rounded,
// 1.18x with both -- the exact-fp32 kernel's 1.19x (profiles/r03_arbiter_split.txt; all nine products with truncation: 1.53x).
#ifndef PINN_SP_ROUND
#define PINN_SP_ROUND 2         // 0: hi and mid by truncation; 1: mid rounded to nearest; 2: hi and mid rounded
#endif
#define PINN_SP_NPROD 6         // partial products per GEMM step: those with i + j <= 2 (all nine bought no accuracy: section 6b)
// bit patterns whose UPPER halves are the three bf16 parts of x (b0: hi, b1: mid, b2: lo); x = hi + mid + lo exactly
PINN_DEVICE void pinn_split3(float x, unsigned& b0, unsigned& b1, unsigned& b2) {
    if (PINN_ABL & 1) { b0 = b1 = b2 = pinn_fbits(x); return; }
    b0 = pinn_fbits(x) + (PINN_SP_ROUND >= 2 ? 0x8000u : 0u);
    const float r1 = x - pinn_bitsf(b0 & 0xffff0000u);
    b1 = pinn_fbits(r1) + (PINN_SP_ROUND >= 1 ? 0x8000u : 0u);
    const float r2 = r1 - pinn_bitsf(b1 & 0xffff0000u);
    b2 = pinn_fbits(r2);
}
struct PinnSplit4 { pinn_u32x2 hi, mid, lo; };            // four consecutive units of one (point, stream): 8 bytes per plane
PINN_DEVICE PinnSplit4 pinn_split4(f32x4 v) {
    unsigned b0[4], b1[4], b2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pinn_split3(v[r], b0[r], b1[r], b2[r]);
    PinnSplit4 o;
    o.hi = pinn_u32x2{pinn_pack_hi16(b0[0], b0[1]), pinn_pack_hi16(b0[2], b0[3])};
    o.mid = pinn_u32x2{pinn_pack_hi16(b1[0], b1[1]), pinn_pack_hi16(b1[2], b1[3])};
    o.lo = pinn_u32x2{pinn_pack_hi16(b2[0], b2[1]), pinn_pack_hi16(b2[2], b2[3])};
    return o;
}
// byte offset, inside one plane, of the 16-byte chunk `chunk` (8 units) of row `row`: rows of ROWB bytes, chunk index swizzled
template <int ROWB>
PINN_DEVICE int pinn_sp_off(int row, int chunk) {
    // (rows of 128 bytes hold 8 chunks: XOR with row & 7; longer rows XOR the low four chunk bits with row & 15, so that the
    //  16 rows a ds_read_b128 lane group touches land in 16 different 16-byte slots of the 256-byte bank row)
    constexpr int M = (ROWB / 16 >= 16) ? 15 : 7;
    return row * ROWB + ((chunk ^ (row & M)) << 4);
}
// the six products of a split-bf16 GEMM step that matter (a_i b_j with i + j <= 2; the three dropped ones are below
// 2^-24 of |a b|), SMALL TERMS FIRST: measured against fp64, tools/ubench/split_bf16.cpp (K = 64: max error 9e-8 of
// sum |a b| against 2e-7 for the exact-fp32 MFMA chain; large terms first: 3.6e-7)
PINN_DEVICE f32x4 pinn_mfma_split6(const pinn_s16x8 (&a)[3], const pinn_s16x8 (&b)[3], f32x4 c) {
    c = pinn_mfma16_bf16(a[2], b[0], c);
    c = pinn_mfma16_bf16(a[0], b[2], c);
    c = pinn_mfma16_bf16(a[1], b[1], c);
    c = pinn_mfma16_bf16(a[1], b[0], c);
    c = pinn_mfma16_bf16(a[0], b[1], c);
    c = pinn_mfma16_bf16(a[0], b[0], c);
    return c;
}
// the four components of v summed over the 16 lanes of a DPP row, step-major (four independent adds per DPP step: a lone
// chain pays two wait states between its dependent steps; pinn_port.h)
// (batching the four DPP chains step-major, pinn_row_sum16_n, measured no difference on the tile kernels: the second wave per SIMD
//  already covers the wait states -- DESIGN.md section 6a)
PINN_DEVICE f32x4 pinn_row_sum16_v4(f32x4 v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = pinn_row_sum16(v[r]);
    return v;
}

#if defined(PINN_FIT_PROF) && !defined(PINN_EMU)
static __device__ long long g_pinn_fitprof[8];
#endif
// VAR (experiment bits): 1 = accumulate dW in the partial buffer although the layer count is static, 2 = two waves per
// SIMD (two workgroups per CU), 4 = no whole-layer weight prefetch; 8 = layout breadth: Sin / identity activations and
// skip connections ('R ... +' layouts; the skipped activations ride in registers through the forward half and in extra
// slab slots through the reverse half); 16, 32, 48 = shape facts of a common training step (PinnShape 1, 2, 3) fixed at compile time;
// 256 = two TEAMS: one 8-wave workgroup runs two independent tile streams (team = waves 0-3 / 4-7, each with its own LDS
// activation buffers, accumulators and slab) that share the read-only staging of W^T in LDS and end in ONE partial row --
// the second wave per SIMD of the two-workgroups-per-CU form (VAR 2) without its prices: no transposed global weight copy
// (a launch per step), no second partial row per CU in the reduction. Shape-specialised, static-depth kernels only.
// 128 = WGX: the weight gradients of the hidden->hidden layers are NOT accumulated here -- the kernel streams gz_a and the
// saved jets of every tile to HBM (lane-private, coalesced) and pinn_wgrad_kernel (pinn_wgrad_kernel.h) turns them into dW
// with the whole HP x HP accumulator in registers (widths >= 128, where a workgroup's dW does not fit on chip).
// 64 = SLABL: the saved jets of a shape-specialised static-depth kernel in LDS instead of the global slab (measured slower, off).
// 512 = SPLIT: the hidden-layer GEMMs on v_mfma_f32_16x16x32_bf16 with every fp32 operand split exactly into three bf16 (round 3;
// pinn_set_gemm_mode). 1024 (with 8) = the breadth kernel for skip connections over Tanh / Sigmoid layers only:
// one-bit activation codes, no skips that start in front of an activation.
// VW > 1 (round 5, pinn_fit_kernel.h only): VW VIRTUAL workgroups in one hardware workgroup -- each with its own LDS block, slab, tile
// stream and partial row, exactly the workgroups of a grid of VW (times the hardware grid), but on ONE CU, so that whatever follows the
// pass (the sum over the partial rows, Adam) needs a workgroup barrier and not a device-scope one. Narrow nets only (one or two waves
// per virtual workgroup); the barriers inside the tile loop become wave-local (one wave: LDS executes a wave's accesses in order) or an
// arrival counter in the virtual workgroup's own LDS block (two waves), so the virtual workgroups run their tiles independently; the
// barriers in front of and behind the loop stay hardware barriers (every virtual workgroup passes them once per call).
template <int HP, int ND, int N2, int MT, int LHC, int ACTC, bool COMB = false, int VAR = 0, int VW = 1>
// occupancy hint: the fused kernel of a 64/128-wide net wants the whole register file of a SIMD (one wave per SIMD,
// no spills); narrower nets (1-2 waves per workgroup) run several workgroups per CU
#ifndef PINN_WAVES_PER_SIMD
#define PINN_WAVES_PER_SIMD 1
#endif
PINN_DEVICE void pinn_tile_body(const PinnKArgs& A, const float* params_, float* partials_, const float* xs_in, float* aux_in, int rstride_in) {
    // params_ / partials_ / xs_ / aux_: the parameter buffer the pass reads, the block of partial gradient rows it writes, the batch and the
    // rows of the x-only pre-pass -- A.params / A.partials / A.xs / A.aux for an ordinary launch; the one-launch fit chunk hands over the
    // workgroup's own copies (the one-CU form: in LDS); rstride_: the row stride of partials_ (A.p_core / rounded up to 16 bytes)
    // (the body of the tile kernel as a function: pinn_tile_kernel below runs it once per launch, pinn_fit_kernel.h -- a whole chunk of
    //  fit iterations in one launch, round 5 -- once per iteration)
    using C = PinnCfg<HP, ND, N2, MT, (VAR & 512) != 0>;
    constexpr int S = C::S, NT = C::NT, NTW = C::NTW, NW = C::NW, T = C::T, LDA = C::LDA, NTHREADS = C::NTHREADS;
    // (only the one-CU fit chunk has copies of its own: everybody else reads the launch arguments where they are needed -- as values
    //  carried through the whole kernel the three cost BASELINE config 2's kernel 0.8 %, same-box A/B)
    const float* xs_ = (VW > 1) ? xs_in : A.xs;
    float* aux_ = (VW > 1) ? aux_in : A.aux;
    const int rstride_ = (VW > 1) ? rstride_in : A.p_core;
    constexpr bool DWG = (LHC < 0) || (VAR & 1), ONEBUF = C::ONEBUF, SKIPS = (VAR & 8) != 0;
    constexpr int LHREG = DWG ? 1 : PINN_LHMAX;            // layers with register-resident dW accumulators
    // VAR 64: saved jets in LDS instead of the global slab (no slab traffic at all); W^T then lives in the workgroup's own
    // global scratch, written in the prologue (the LDS it used to occupy is what the jets need)
    constexpr bool SLABL = (VAR & 64) != 0;
    constexpr bool WGX = (VAR & 128) != 0;
    constexpr bool TEAMS2 = (VAR & 256) != 0;
    constexpr bool VWG = VW > 1;                           // virtual workgroups (above): `team` is the virtual workgroup's index
    constexpr int TEAMS = TEAMS2 ? 2 : VW;
    static_assert(!VWG || (!TEAMS2 && !(VAR & (2 | 64 | 128 | 512)) && NW <= 2 && HP <= 32), "virtual workgroups: the narrow nets' plain kernels");
    // VAR 512: split-bf16 GEMMs. Every hidden-layer GEMM (forward, data gradient, weight gradient) runs on
    // v_mfma_f32_16x16x32_bf16 with both operands split exactly into hi + mid + lo bf16 and the six products a_i b_j,
    // i + j <= 2, accumulated in fp32 (pinn_mfma_split6): 6 x 16 cycles of the matrix pipe per K = 32 instead of 8 x 32 cycles
    // of the fp32 vector lanes, and the jets' VALU work co-executes. The activations cross the LDS as three bf16 planes
    // (PinnCfg::SP_*), split where they are written; the weight fragments come pre-split from global memory (A.wsp,
    // pinn_wsplit_kernel, L2-resident); the weight-gradient operands are read point-contiguous out of the same planes with
    // ds_read_b64_tr_b16. Accumulator layouts, jets, point stage, slab and reductions are those of the exact kernel.
    constexpr bool SPLIT = C::SPLIT;
    // (VAR 2, two INDEPENDENT workgroups per CU, stays refused for the split kernels -- not for the reason round 3 gave (scratch: the
    //  failing kernel has no spill at all), but for the one round 4 found ("run-to-run different gradients"):
    //  with the SLP vectoriser's packed fp32 code (v_pk_fma_f32 on register pairs gathered by v_mov_b32) a wave whose SIMD partner is
    //  in a bf16-MFMA phase AT THE SAME TIME now and then computes a wrong low half in lanes 48-63 of a packed instruction (first layer:
    //  the term W1[n][1] * y of one unit pair, inputs verified bit-identical, tools/diff_runs.py --net) -- run-to-run different, 1e-5 in
    //  the gradients. Two teams in one workgroup share every barrier, so their vector phases never meet the other team's GEMM phases;
    //  two independent workgroups (or flag-synchronised teams, PINN_TEAM_FLAGS) drift. Builds WITHOUT packed fp32 code
    //  (-fno-slp-vectorize) are bit-repeatable in this form too (30 of 30 runs) but 8 % slower than the two-team kernel with it;
    //  experiment builds lift the refusal with -DPINN_VAR2_REFUSED=0, tools/var2.sh)
    // Widths >= 128 (round 3, later): the WGX form -- forward and data-gradient GEMMs here, weight fragments streamed K block by
    // K block (they do not fit the registers), weight gradients in pinn_wgrad_kernel's own split form.
#ifndef PINN_VAR2_REFUSED
#define PINN_VAR2_REFUSED 2     // (0 in the experiment builds that look for the cause of the finding above)
#endif
    static_assert(!SPLIT || (HP == 64 && NTW == 1 && LHC >= 1 && !(VAR & (1 | PINN_VAR2_REFUSED | 8 | 64 | 128)) && ((S * T) % 32) == 0 &&
                             ((T == 16 && S % 2 == 0) || T == 32)) ||
                            (HP >= 128 && (VAR & 128) && !(VAR & (1 | 2 | 8 | 64 | 256))),
                  "split-bf16 kernels: width 64 with static depth and register-resident dW (K = S * T a multiple of 32), or the WGX kernels of widths >= 128");
    constexpr bool SPW = SPLIT && HP >= 128;               // wide form: weights streamed inside the GEMM
    constexpr int SPK = (SPLIT && !SPW) ? C::SP_KB : 1;    // K blocks whose weight fragments a wave holds at once (width 64: all)
    static_assert(!TEAMS2 || (((VAR >> 4) & 3) != 0 && LHC >= 1 && !(VAR & (1 | 2 | 8 | 64 | 128)) && NW == 4 && C::wt_fits_teams(LHC)),
                  "two-team kernels: shape-specialised, static depth, 4 waves per team, W^T of all layers in LDS");
    constexpr bool SNT = HP >= PINN_SLAB_NT_MIN_HP;        // streaming (non-temporal) slab stores
    static_assert(!WGX || (DWG && !SLABL), "WGX kernels: generic depth, global slab");
    static_assert(!SLABL || (((VAR >> 4) & 3) != 0 && LHC >= 1 && !(VAR & 8) && !(VAR & 1) && C::slabl_fits(LHC)),
                  "slab-in-LDS kernels: shape-specialised, static depth, no skips, and the jets must fit");
    constexpr bool WTL = !SPLIT && ((C::wt_fits(LHC) && !SLABL && !(VAR & 2)) || TEAMS2);        // transposed hidden weights staged in LDS
    // (virtual workgroups: C::wt_fits(LHC) of ONE block is the launcher's condition for the shared copy as well -- pinn_inst.inc sizes
    //  VW so that VW blocks + W^T fit)
    // widths >= 128: the data-gradient A operand comes from a transposed copy of the weights in global memory (one b128 per
    // K quad like the forward GEMM) instead of four strided global_load_dword per quad
    // (VAR 2, two workgroups per CU: W^T of both does not fit the LDS beside the activation buffers)
    constexpr bool WTG = !SPLIT && (C::WTG || SLABL || (VAR & 2) != 0);
    constexpr int SPEC = (VAR >> 4) & 3;                   // VAR 16/32/48: training shape 1/2/3 fixed at compile time
    // VAR 2048 (round 6): the shape-specialised two-team kernel with a residual PROGRAM (PinnShape: SPEC | 4) -- its program registers
    // (PINN_MAX_REGS values + adjoints per point of a team's tile) live in the rows of the team's bias-gradient block that a static-depth
    // net never uses (rows 8 .. PINN_MAX_LAYERS - 1 of `accB`: the team blocks of the LDS carve end in front of the program registers,
    // and with W^T behind them the 160 KB are used up); the point stage runs on the first T threads of the team (every lane running it,
    // PTALL, would have sixteen replicas add into the same adjoint registers)
    constexpr bool PROG = (VAR & 2048) != 0;
    constexpr int SPECP = SPEC | (PROG ? 4 : 0);
    using SH = PinnShape<SPECP, ND>;
    static_assert(!PROG || (TEAMS2 && SPEC != 0 && LHC >= 1 && LHC + 1 <= 8 && !SPLIT &&
                            2 * PINN_MAX_REGS * T <= (C::ACCB_ROWS - 8) * HP),
                  "program residuals on the two-team kernels: static depth, registers inside the unused bias-gradient rows");
    // two teams: everything below is written in TEAM-local terms (tid, wave, LDS block, virtual block index); the teams meet
    // at the barriers only (same trip counts by construction) and in the shared W^T
    // (the team index is wave-uniform -- NTHREADS is a multiple of 64 -- and the compiler should know: tile index, LDS block, slab and
    //  partial row derived from it then live in scalar registers instead of 64-bit vector pairs. Round 5: the two-team kernel of BASELINE
    //  config 2 reloaded two such pairs from scratch in the middle of every tile)
    const int gtid = PINN_TID, team = (TEAMS2 || VWG) ? pinn_wave_uniform(gtid / NTHREADS) : 0;
    const int tid = (TEAMS2 || VWG) ? gtid % NTHREADS : gtid, lane = tid & 63, wave = tid >> 6;
    const int vbid = PINN_BID * TEAMS + team, vnblk = PINN_NBLK * TEAMS;
    const int rowid = VWG ? vbid : PINN_BID;               // partial row (two teams share their workgroup's, virtual workgroups own one each)
    const int lr = lane & 15, lq = lane >> 4;
    const int lh = (LHC >= 0) ? LHC : A.lh;
    // activation of index a (0: first layer ... lh: last hidden layer)
    // (plain instantiations only know tanh / sigmoid -- one bit, which lets the compiler drop the sin / identity paths;
    //  the full set runs on the VAR 8 instantiations, see the launcher)
    // (VAR 8 | 1024: skip connections over Tanh / Sigmoid layers only -- the usual residual PINN -- keep the one-bit code: with the
    //  sin / softplus / SiLU / GELU paths compiled in, the width-128 breadth kernel spills 283 registers and runs 8 % slower)
    // skips that START in front of an activation ('fRa'): the generic-depth full breadth kernels only (the selects and the guarded slab
    // reads cost the static-depth Sin kernel 3.5 % -- 9 % with the guard inside the jet loop -- and it never sees a skip)
    constexpr bool SRCPRE = SKIPS && !(VAR & 1024) && LHC < 0;
    // nested skips (round 5): the same kernels. ONE skip rides in registers (`hskip`); a skip during whose life another one opens
    // ("outer", A.skip_outer) parks its jets in its own slab slot at 'R' and reads them back at '+' -- lane-private, written and
    // read by the same lane, L2-resident -- so that nesting costs no second register set (the launcher sends nested nets here)
    constexpr bool NEST = SRCPRE && ACTC == -2;
    // (full breadth kernels come in two sets, round 5: ACTC -1 knows the activation codes 0 .. 7 -- the seven of round 4 and ReLU -- and
    //  no nested skips: exactly the kernels round 4 measured; ACTC -2 knows all sixteen codes and parks the outer skip of a nest. The
    //  second set spills 50 - 150 registers more at width 256, which the nets of the first set should not pay)
    constexpr bool ALLACT = ACTC == -2;
    // tanh by its minimax polynomial below |z| = 0.45 (pinn_act): an instantiation of its own, ACTC = PINN_ACT_TANH | PINN_ACT_TANH_POLYBIT,
    // picked by pinn_set_tanh_mode(PINN_TANH_ACCURATE). Same-box A/B on BASELINE config 2 (round 5, profiles/r05_headline_ab.txt): +2.5 % kernel
    // time (the select per value costs the forward epilogues their packed fp32 code) for a trained-state gradient error of 0.6 - 1.2x the fp32
    // reference's own instead of 1.7 - 1.9x (bench.py `parity_trained_state`). Not the default: the default form already meets the survey's bar.
    constexpr int TPOLY = (ACTC >= 0 && (ACTC & PINN_ACT_TANH_POLYBIT)) ? PINN_ACT_TANH_POLYBIT : 0;
    constexpr int ACTK = (ACTC >= 0) ? (ACTC & 0xff) : ACTC;          // the activation code proper
    // (the parameter of a configured LeakyReLU / ELU / Softplus travels with the code in the full breadth kernels -- round 6)
    auto act_at = [&](int a) -> PinnAct {
        if (ACTC >= 0) return PinnAct(ACTK);
        constexpr bool FULL = SKIPS && !(VAR & 1024);
        return PinnAct(pinn_act_code(A.act_codes, a) & (FULL ? (ALLACT ? 15 : 7) : 1), FULL ? A.act_par[a] : 0.0f);
    };
    auto with_poly = [&](PinnAct a) -> PinnAct { return PinnAct(a.c | TPOLY, a.p); };
    // skip connection ending / starting at activation a (or -1); slab slot of skip k
    auto skip_into = [&](int a) -> int {
        int k = -1;
        if (SKIPS) for (int i = 0; i < A.n_skips; ++i) if (A.skip_dst[i] == a) k = i;
        return k;
    };
    auto skip_from = [&](int a) -> int {
        int k = -1;
        if (SKIPS) for (int i = 0; i < A.n_skips; ++i) if (A.skip_src[i] == a) k = i;
        return k;
    };
    const float* cw = A.comb_w;
    const int d = SH::d(A);
    const bool train = SH::mode(A) != PINN_MODE_FORWARD;

    PINN_SMEM(smem_all);
    // (virtual workgroups of a kernel with program registers carry them in their block; W^T sits once behind all blocks)
    constexpr int TEAM_FLOATS = (VWG && ((VAR >> 4) & 3) == 0) ? C::SMEM_FLOATS : C::TEAM_FLOATS;
    float* smem = smem_all + team * TEAM_FLOATS;               // (one team: the whole block)
    float* xs_base = smem + C::O_XS;
    float* W1s = smem + C::O_W1;
    float* b1s = smem + C::O_B1;
    float* WLs = smem + C::O_WL;
    float* bufA = smem + C::O_BUFA;
    float* bufB = smem + C::O_BUFB;
    float* netp = smem + C::O_NET;
    float* gnetb = smem + C::O_GNET;
    float* accB = smem + C::O_ACCB;
    float* accW1 = smem + C::O_ACCW1;
    float* scal = smem + C::O_SCAL;
    int* tbar = reinterpret_cast<int*>(smem + C::O_TBAR);
    float* pregs = PROG ? accB + 8 * HP : smem + C::O_PREG;
    float* padj = PROG ? accB + 8 * HP + PINN_MAX_REGS * T : smem + C::O_PADJ;       // (zeroed with accB below)

    // -DPINN_FIT_PROF (experiment builds, pinn_fit_kernel.h): clock ticks of thread 0 between marks of this function
#if defined(PINN_FIT_PROF) && !defined(PINN_EMU)
    long long fpb_last = __builtin_readcyclecounter();
#define FPB(i) if (PINN_TID == 0 && PINN_BID == 0) { const long long fpb_now = __builtin_readcyclecounter(); g_pinn_fitprof[i] += fpb_now - fpb_last; fpb_last = fpb_now; }
#else
#define FPB(i)
#endif
    // ---- one-time staging of the small layers and zeroing of the LDS accumulators -------------------------------
    for (int i = tid; i < HP * PINN_XS_LD; i += NTHREADS) {
        const int n = i / PINN_XS_LD, c = i % PINN_XS_LD;
        W1s[i] = (c < d) ? params_[n * d + c] : 0.0f;
        accW1[i] = 0.0f;
    }
    for (int i = tid; i < HP; i += NTHREADS) { b1s[i] = params_[A.off_b1 + i]; WLs[i] = params_[A.off_wl + i]; }
    if (tid == 0) tbar[0] = 0;
    for (int i = tid; i < C::ACCB_ROWS * HP; i += NTHREADS) accB[i] = 0.0f;
    float* WTs = (TEAMS2 || VWG) ? smem_all + TEAMS * TEAM_FLOATS : smem + C::O_WT;
    const float* wtg = A.wt + (SLABL ? (size_t)PINN_BID * (size_t)lh * HP * HP : (size_t)0);
    if (SLABL && train) {
        // wt[l][k][n] = W_l[n][k] in this workgroup's scratch: coalesced reads along k, strided fire-and-forget writes;
        // first read long after the barriers of the first tile's forward half
        float* wtw = const_cast<float*>(wtg);
        for (int i = tid; i < lh * HP * HP; i += NTHREADS) {
            const int l = i / (HP * HP), n = (i / HP) % HP, k = i % HP;
            wtw[((size_t)l * HP + k) * HP + n] = params_[A.off_wh + (size_t)l * A.hidden_stride + n * HP + k];
        }
    }
    // WTs[l][k][n] = W_l[n][k]: coalesced global reads along k, one-time strided LDS writes. ALL loads of a thread are
    // issued before the first LDS write (the registers are free here): the weights were last written by another
    // launch's Adam on other XCDs, so every batch of loads pays a full L2-miss round trip -- one instead of six or more
    // (prologue phase counters: 9.4 K -> cycles of one round trip on cfg2). (Holding the LDS writes back until the x-only pre-pass
    // below is through, so that the round trip runs behind it, measured +0.8 % / +-0 with Adam in the loop: not kept.)
    constexpr int WT_TOTAL = (LHC > 0 ? LHC : 0) * HP * HP;
    constexpr int NTH_ALL = NTHREADS * TEAMS;               // both teams stage the shared copy together
    constexpr int WT_PER = (WT_TOTAL + NTH_ALL - 1) / NTH_ALL;
    constexpr int WT_STAGE_BATCH = 64;                      // staging loads issued before the first LDS write
    constexpr int WT_B = WT_PER < 1 ? 1 : (WT_PER < WT_STAGE_BATCH ? WT_PER : WT_STAGE_BATCH);
    auto wt_write = [&](int e0, const float* wreg) {
#pragma unroll
        for (int e = 0; e < WT_B; ++e) {
            const int i = gtid + (e0 + e) * NTH_ALL;
            const int l = i / (HP * HP), n = (i / HP) % HP, k = i % HP;
            if (i < WT_TOTAL) WTs[(l * HP + k) * C::WT_LD + n] = wreg[e];
        }
    };
    if (WTL && train) {
        for (int e0 = 0; e0 < WT_PER; e0 += WT_B) {
            float wreg[WT_B];
#pragma unroll
            for (int e = 0; e < WT_B; ++e) {
                const int i = gtid + (e0 + e) * NTH_ALL;
                const int l = i / (HP * HP), n = (i / HP) % HP, k = i % HP;
                wreg[e] = (i < WT_TOTAL) ? params_[A.off_wh + (size_t)l * A.hidden_stride + n * HP + k] : 0.0f;
            }
            wt_write(e0, wreg);
        }
    }
    if (!SLABL && !TEAMS2 && !(VWG && ((VAR >> 4) & 3) != 0))
        for (int i = tid; i < PINN_MAX_REGS * T; i += NTHREADS) padj[i] = 0.0f;   // (team blocks end before the program registers)
    const float bL = params_[A.off_bl];

    // persistent per-lane accumulators
    f32x4 dW[LHREG][DWG ? 1 : NT][NTW];
#pragma unroll
    for (int l = 0; l < LHREG; ++l)
#pragma unroll
        for (int o = 0; o < (DWG ? 1 : NT); ++o)
#pragma unroll
            for (int j = 0; j < NTW; ++j) dW[l][o][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this lane's element of weight-gradient tile (o, j) of hidden layer li inside the workgroup's partial buffer
    auto dwg_ptr = [&](int li, int o, int j, int r) -> float* {
        return partials_ + (size_t)rowid * rstride_ + A.off_wh + (size_t)li * A.hidden_stride +
               (o * 16 + lq * 4 + r) * HP + (wave * NTW + j) * 16 + lr;
    };
    if (DWG && !WGX && train) {
        for (int li = 0; li < lh; ++li)
            for (int o = 0; o < NT; ++o)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) *dwg_ptr(li, o, j, r) = 0.0f;
    }
    f32x4 accWL[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) accWL[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fast instantiations: the bias gradients (and the first W1R columns of dW1) are summed per LANE in registers over
    // all tiles and reduced over the 16 points of a lane row ONCE at the end, instead of 4 DPP adds per value plus an
    // LDS read-modify-write in every tile
    // (only where the jets leave register head-room: the S = 4, 32-point kernel of cfg2 uses all 512 registers and
    //  started to spill with these 32 more -- measured +3.7 % time there, -4.8 % on the S = 2 kernel of cfg4)
#ifndef PINN_REGB_MAX
#define PINN_REGB_MAX 6
#endif
#ifndef PINN_REGB_MAX_SPEC
#define PINN_REGB_MAX_SPEC 8
#endif
    // (register accumulators for the bias / first-layer gradients in the two-streams-per-CU kernels as well: 80-160 B per lane of
    //  scratch, slower -- DESIGN.md section 6a)
    constexpr bool REGB = !DWG && !(VAR & (2 | 256)) && (S * MT * NTW <= (SPEC != 0 ? PINN_REGB_MAX_SPEC : PINN_REGB_MAX));
    // (two workgroups per CU live on 256 registers: accumulators for the layers that exist and two input columns only)
    constexpr int W1R = (VAR & (2 | 256)) ? 2 : 4;
    constexpr int NBR = (VAR & (2 | 256)) && LHC >= 0 ? LHC + 1 : PINN_LHMAX + 1;
    f32x4 accBr[REGB ? NBR : 1][NTW], accW1r[REGB ? W1R : 1][NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
#pragma unroll
        for (int a = 0; a < (REGB ? NBR : 1); ++a) accBr[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < (REGB ? W1R : 1); ++c) accW1r[c][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float sum_loss = 0.0f, sum_ls = 0.0f, sum_bl = 0.0f, sum_ic = 0.0f;

    f32x4* slab = SLABL ? reinterpret_cast<f32x4*>(smem + C::O_PREG)
                        : (A.slab ? A.slab + (size_t)vbid * C::slab_vec4_per_wg(lh, SKIPS ? A.n_skips : 0) : nullptr);
    f32x4* gzs = nullptr;            // WGX: this tile's block of A.gzslab (set per tile, like `slab`)
    auto gz_at = [&](int a, int s, int j, int mt) -> f32x4* {      // a = 1 .. lh
        return gzs + (((size_t)((a - 1) * S + s) * NTW + j) * MT + mt) * NTHREADS + tid;
    };
    auto slab_at = [&](int a, int s, int j, int mt) -> f32x4* {
        // SLABL: compact slots -- activation 0 keeps its value only, the top activation never comes here
        const size_t slot = SLABL ? (size_t)(a == 0 ? 0 : 1 + (a - 1) * S + s) : (size_t)a * S + s;
        return slab + ((slot * NTW + j) * MT + mt) * NTHREADS + tid;
    };
    // slab slot group of skip k: the carried activations (forward) / the gradient on its way back to the source (reverse). One
    // slot serves both in the kernels that accumulate dW themselves (the activations are consumed first); the WGX kernels keep
    // the activations for pinn_wgrad_kernel (h_{a-1} of the layer behind a '+') and send the gradient through a slot of its own
    auto skip_grad_slot = [&](int k) -> int { return lh + 1 + ((WGX && SKIPS) ? A.n_skips : 0) + k; };
    auto unit0 = [&](int j) { return (wave * NTW + j) * 16 + 4 * lq; };     // first of this lane's 4 units
    // outer skip of a nest: its carried jets wait in the skip's slab slot (the slot the reverse half / pinn_wgrad_kernel reads anyway)
    auto skip_park = [&](int k, int s, int j, int mt, f32x4 v) {
        if (WGX) pinn_st4_stream<HP >= PINN_SLAB_NT_MIN_HP>(slab_at(lh + 1 + k, s, j, mt), v);
        else *slab_at(lh + 1 + k, s, j, mt) = v;
    };
#if defined(PINN_DUMP_NET) && !defined(PINN_EMU)
    // experiment builds: a checksum per (tile, layer, wave) of the value stream a wave has just produced, behind the point dump
    // (rows 8.. of A.prof as floats): which layer and which wave of a tile differs first between two runs?
    auto dump_layer = [&](long long tile_, int layer, const f32x4& v) {
        if (!A.prof) return;
        float t = fabsf(v[0]) + fabsf(v[1]) + fabsf(v[2]) + fabsf(v[3]);
        t = pinn_rows_sum(pinn_row_sum16(t));
        if (lane == 0) reinterpret_cast<float*>(A.prof)[8LL * A.n_points + ((tile_ * 16 + layer) * 4 + wave)] = t;
    };
#endif
    // split-bf16 kernels: four consecutive units [n0, n0 + 4) of row `row` (= s * T + point) into the three planes of `buf`
    auto sp_store = [&](float* buf, int row, int n0, f32x4 v) {
        char* b = reinterpret_cast<char*>(buf) + pinn_sp_off<C::SP_ROW_BYTES>(row, n0 >> 3) + ((n0 >> 2) & 1) * 8;
        const PinnSplit4 sp = pinn_split4(v);
#ifndef PINN_EMU
        if (PINN_ABL & 8) { asm volatile("" :: "v"(sp.hi), "v"(sp.mid), "v"(sp.lo), "v"(b)); return; }
#endif
        *reinterpret_cast<pinn_u32x2*>(b) = sp.hi;
        *reinterpret_cast<pinn_u32x2*>(b + C::SP_PLANE_BYTES) = sp.mid;
        *reinterpret_cast<pinn_u32x2*>(b + 2 * C::SP_PLANE_BYTES) = sp.lo;
    };
    // ... and the fragment of 8 units (chunk = K block * 4 + lq) of row `row`, all three planes: the B operand of a GEMM step
    auto sp_frag = [&](const float* buf, int row, int chunk, pinn_s16x8 (&f)[3]) {
        const char* b = reinterpret_cast<const char*>(buf) + pinn_sp_off<C::SP_ROW_BYTES>(row, chunk);
#pragma unroll
        for (int p = 0; p < 3; ++p) f[p] = *reinterpret_cast<const pinn_s16x8*>(b + p * C::SP_PLANE_BYTES);
    };
    // weight fragments of hidden layer l, direction dir (0: W for the forward GEMM, 1: W^T for the data gradient), this wave's tile
    // (buffer loads: ONE vector register of offsets, lane * 16, for all 36 fragments of the net, the fragment's offset in a scalar
    //  register -- with plain pointers hipcc hoists a 64-bit VGPR address pair per fragment out of the tile loop, spills them, and
    //  every load then waits for the scratch reload of its own address)
    const PinnRows wsp_rows = pinn_rows(A.wsp, SPLIT ? (unsigned)C::wsp_bytes(lh > 0 ? lh : 1) : 0u);
    const int wave_s = pinn_wave_uniform(wave);
    // fragment (l, dir, kb, tile j of this wave, plane p)
    auto sp_wfrag = [&](int l, int dir, int kb, int j, int p) -> pinn_s16x8 {
        return __builtin_bit_cast(pinn_s16x8, pinn_rows_ld4(wsp_rows, lane * 16, (int)(C::wsp_frag(l, dir, kb, 0, p) * 16) + (wave_s * NTW + j) * 3 * 1024));
    };
    auto sp_weights = [&](int l, int dir, pinn_s16x8 (&w)[SPK][3]) {
        if constexpr (SPLIT && !SPW) {
#pragma unroll
            for (int kb = 0; kb < SPK; ++kb)
#pragma unroll
                for (int p = 0; p < 3; ++p) w[kb][p] = sp_wfrag(l, dir, kb, 0, p);
        }
    };
    // out^T[16 units of this wave][points] += W-fragments . rows of `bbuf` (forward: h_{l-1}, data gradient: gz_a), all S * MT
    // (row tile, stream) rows of the tile: two rows per step, their B fragments (three planes each) fetched one step ahead,
    // the twelve MFMAs of a step alternating between the two accumulators, small products first (pinn_mfma_split6's order)
#ifndef PINN_SP_PIPE
#define PINN_SP_PIPE 1          // split-bf16 GEMMs: operand fragments of step n + 1 in flight during the MFMAs of step n (0: fetched at the step's start)
#endif
#ifndef PINN_SP_PIPE_W
#define PINN_SP_PIPE_W PINN_SP_PIPE     // ... of the weight-gradient GEMM
#endif
#define PINN_SP_G 2             // (row tile, stream) rows per step of the forward / data-gradient GEMMs
#define PINN_SP_OG 2            // output tile rows per step of the weight-gradient GEMM
    auto sp_gemm = [&](const float* bbuf, const pinn_s16x8 (&w)[SPK][3], f32x4 (&out)[NTW][MT][S]) {
        constexpr int NR = MT * S, G = (NR % PINN_SP_G == 0) ? PINN_SP_G : 1, NG = NR / G, STEPS = C::SP_KB * NG;
        constexpr int NB = PINN_SP_PIPE ? 2 : 1;
        constexpr int pa[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, pb[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};     // (a part, b part), small products first
        pinn_s16x8 bf[NB][G][3];
        auto load = [&](int step, pinn_s16x8 (&f)[G][3]) {
            const int kb = step / NG, g0 = (step % NG) * G;
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int r = g0 + i, mt = r / S, sidx = r % S;
                sp_frag(bbuf, sidx * T + mt * 16 + lr, 4 * kb + lq, f[i]);
            }
        };
        if (PINN_SP_PIPE) load(0, bf[0]);
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            PINN_SCHED_BARRIER();
            if (PINN_SP_PIPE) { if (step + 1 < STEPS) load(step + 1, bf[(step + 1) % NB]); }
            else load(step, bf[0]);
            if (!PINN_SCHED_IL) PINN_SCHED_BARRIER();
            const int kb = step / NG, g0 = (step % NG) * G;
#pragma unroll
            for (int t = 9 - PINN_SP_NPROD; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int r = g0 + i;
                    out[0][r / S][r % S] = pinn_mfma16_bf16(w[SPK > 1 ? kb : 0][pa[t]], bf[step % NB][i][pb[t]], out[0][r / S][r % S]);
                }
            if (PINN_SP_PIPE && step + 1 < STEPS) pinn_sched_interleave<6 * G, 3 * G>();
            PINN_SCHED_BARRIER();
        }
    };
    // the same GEMM for widths >= 128 (SPW): NTW output tiles per wave, the weight fragments of a K block (NTW x three planes)
    // arrive from global memory / L2 one K block ahead (all K blocks of a layer do not fit the registers: 2 x 8 x 3 fragments at
    // width 256), the B fragments of a step's rows one step ahead
    auto sp_gemm_wide = [&](const float* bbuf, int l, int dir, f32x4 (&out)[NTW][MT][S]) {
        constexpr int NR = MT * S, G = (NR % 2 == 0) ? 2 : 1, NG = NR / G, KB = C::SP_KB, STEPS = KB * NG;
        constexpr int pa[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, pb[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
        pinn_s16x8 wa[2][NTW][3], bf[2][G][3];
        auto load_w = [&](int kb, pinn_s16x8 (&w)[NTW][3]) {
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) w[j][p] = sp_wfrag(l, dir, kb, j, p);
        };
        auto load_b = [&](int step, pinn_s16x8 (&f)[G][3]) {
            const int kb = step / NG, g0 = (step % NG) * G;
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int r = g0 + i, mt = r / S, sidx = r % S;
                sp_frag(bbuf, sidx * T + mt * 16 + lr, 4 * kb + lq, f[i]);
            }
        };
        load_w(0, wa[0]);
        load_b(0, bf[0]);
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            const int kb = step / NG, g0 = (step % NG) * G;
            PINN_SCHED_BARRIER();
            if (step % NG == 0 && kb + 1 < KB) load_w(kb + 1, wa[(kb + 1) & 1]);
            if (step + 1 < STEPS) load_b(step + 1, bf[(step + 1) & 1]);
            if (!PINN_SCHED_IL) PINN_SCHED_BARRIER();
#pragma unroll
            for (int t = 9 - PINN_SP_NPROD; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < G; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        const int r = g0 + i;
                        out[j][r / S][r % S] = pinn_mfma16_bf16(wa[kb & 1][j][pa[t]], bf[step & 1][i][pb[t]], out[j][r / S][r % S]);
                    }
            if (step + 1 < STEPS) pinn_sched_interleave<PINN_SP_NPROD * G * NTW, 3 * G>();
            PINN_SCHED_BARRIER();
        }
    };
    // weight-gradient GEMM: MFMA k-slot (lq, m) -> point of the tile. Any bijection works (K is a sum index); this one
    // puts lanes lq and lq+1 two rows (2*LDA = 16 mod 32 banks) apart, so the ds_read_b32 column reads are conflict-free.
    auto wg_pt = [&](int m) { return 2 * lq + (m & 1) + 8 * (m >> 1); };

    const long long ntiles = A.tile_end;             // tiles [A.tile_begin, A.tile_end) of the batch belong to this launch
    // the points of a tile are fetched one tile ahead into registers (HBM latency hidden behind a whole tile)
    constexpr int NPRE = (T * PINN_XS_LD + NTHREADS - 1) / NTHREADS;
    float xpre[NPRE];
    auto fetch_points = [&](long long tile) {
#pragma unroll
        for (int e = 0; e < NPRE; ++e) {
            const int i = tid + e * NTHREADS;
            const int pt = i / PINN_XS_LD, c = i % PINN_XS_LD;
            const long long g = tile * T + pt;
            xpre[e] = (i < T * PINN_XS_LD && c < d && tile < ntiles && g < A.n_points) ? xs_[g * d + c] : 0.0f;
        }
    };
    auto store_points = [&](float* dst) {
#pragma unroll
        for (int e = 0; e < NPRE; ++e) {
            const int i = tid + e * NTHREADS;
            if (i < T * PINN_XS_LD) dst[i] = xpre[e];
        }
    };
    // the points live in a double-buffered LDS tile: tile k reads buffer k&1 while the points of tile k+1 are written
    // to the other one in the middle of tile k (several barriers away from both its last reader and its first reader),
    // so neither the staging nor the end of a tile needs a barrier of its own
    FPB(0)
    if (A.pre.n_ops > 0) {
        // x-only pre-pass (source terms, variable coefficients) for the points of this workgroup's own tiles, all threads,
        // NTHREADS / T tiles per sweep; the rows land in A.aux and are read back (by the point-stage threads of the same
        // workgroup, hence the fence + the barrier below) at the top of each tile
        // (its registers live in the activation buffers, which nothing uses before the first tile: as many points per sweep as their
        //  registers fit -- a tile's worth always does)
        // (round 6: DOUBLE registers -- two floats each; the host sends a program whose registers do not fit a tile's worth of points
        //  through the separate launch instead, PinnCfg::PREPASS_FLOATS_PER_LANE / run_train)
        static_assert(C::O_BUFA % 2 == 0 && C::TEAM_FLOATS % 2 == 0 && C::SMEM_FLOATS % 2 == 0, "double registers of the pre-pass: 8-byte aligned");
        int pp_lanes = NTHREADS;
        while (pp_lanes > T && 2 * A.pre_nregs * pp_lanes > C::O_NET - C::O_BUFA) pp_lanes -= T;
        double* pp_regs = reinterpret_cast<double*>(smem + C::O_BUFA) + tid;
        if (tid < pp_lanes) {
            for (long long tile = A.tile_begin + vbid + (long long)(tid / T) * vnblk; tile < ntiles; tile += (long long)(pp_lanes / T) * vnblk) {
                const long long gi = tile * T + tid % T;
                if (gi < A.n_points) pinn_prepass_point(A.pre, A.pre_consts64, xs_ + gi * d, d, aux_, A.n_points, gi, pp_regs, pp_lanes);
            }
        }
        PINN_FENCE_BLOCK();
    }
    FPB(1)
    fetch_points(A.tile_begin + vbid);
    store_points(xs_base);
    fetch_points(A.tile_begin + vbid + vnblk);
    PINN_SYNC();
    PH_DECL
    FPB(2)

    int tile_parity = 0;
    // (two-team kernels with team 1 running one barrier behind team 0, so that its vector phases meet team 0's GEMM phases:
    //  cfg2 -3 %, cfg4 +5 % on the split kernels -- DESIGN.md section 6b; not kept)
    // barriers INSIDE the tile loop. Two-team kernels share nothing between the teams in there (each team its own LDS block and
    // slab; W^T / the split fragments are read-only), so a team's barrier need not hold the other team: PINN_TEAM_FLAGS replaces
    // s_barrier by an arrival counter in the team's LDS block and the teams drift apart -- one team's vector phases then run
    // under the other's GEMM phases on every SIMD they share (they were lock-stepped phase by phase before: section 6b)
#ifndef PINN_TEAM_FLAGS
#define PINN_TEAM_FLAGS 0
#endif
    constexpr bool TEAM_FLAGS = (TEAMS2 && ((PINN_TEAM_FLAGS & (SPLIT ? 1 : 2)) != 0)) || (VWG && NW > 1);
    int tb_round = 0;
    auto tsync = [&]() {
        if constexpr (VWG && NW == 1) {
            PINN_WAVE_SYNC();
        } else if constexpr (TEAM_FLAGS) {
            tb_round += NW;
            pinn_flag_arrive(tbar, lane == 0);
            while (pinn_flag_load(tbar) < tb_round) PINN_SPIN_PAUSE();
            PINN_WAVE_SYNC();
        } else if (!(PINN_ABL & 16)) {
            PINN_SYNC();
        }
    };
    // PINN_TEAM_SKEW (round 6): team 1 of an exact-fp32 two-team kernel runs SKEW barriers behind team 0 -- it passes SKEW empty
    // barriers in front of its first tile, team 0 as many behind its last one, so every hardware barrier still sees all eight waves.
    // With SKEW = half a tile's barriers a team's vector phases (first layer, jet epilogues' tails, point stage, activation reverse +
    // staging) meet the other team's GEMM phases on the SIMD they share instead of its vector phases: the latency of one hides under
    // the MFMA issue of the other (fp32 MFMA and VALU share the issue port -- the sum of both is the floor -- but a wave that waits
    // for LDS, a transcendental or a barrier issues nothing). Not for the split-bf16 kernels (their packed
    // fp32 code must never run under the partner's bf16-MFMA phase).
    // PINN_NOTOPB: no barrier behind the epilogue of the LAST hidden layer -- it writes nothing to LDS (its activations stay in
    // registers), the head dot's partial sums go to `netp`, whose last readers (the previous tile's point stage) are a tile away.
#ifndef PINN_TEAM_SKEW
#define PINN_TEAM_SKEW 0
#endif
    // round 6 micro-structure knobs of the shape-specialised kernels (same-box A/B: profiles/r06_headline_ab.txt):
    // PINN_PT_WAVE_SPREAD  the point-stage threads of team t sit in wave t: the two teams' serial point stages (one thread per point)
    //                      then run on DIFFERENT SIMDs instead of time-slicing SIMD 0, which hosts wave 0 of both teams
    // PINN_GATE_FAST       exp(-log_scale) once per launch, the IC gate's sigmoid on v_exp_f32 / v_rcp_f32
    // PINN_BIAS_EARLY_LD   the LDS read of the bias / first-layer gradient rows issued in front of the arithmetic whose row sums they
    //                      receive (inside the `lr == 0` region the read's round trip sat between the sums and the write)
#ifndef PINN_PT_WAVE_SPREAD
#define PINN_PT_WAVE_SPREAD 0
#endif
#ifndef PINN_GATE_FAST
#define PINN_GATE_FAST 1
#endif
#ifndef PINN_BIAS_EARLY_LD
#define PINN_BIAS_EARLY_LD 0
#endif
#ifndef PINN_NOTOPB
#define PINN_NOTOPB 0
#endif
    // the threads of the serial point stage: lanes [0, T) of wave PTW of the team
    const int ptw = (TEAMS2 && PINN_PT_WAVE_SPREAD) ? team : 0;            // (wave-uniform)
    const int ptid = tid - 64 * ptw;
    const bool pt_thread = ptid >= 0 && ptid < T;
    // exp(-log_scale) of the IC gate, once per launch (PINN_GATE_FAST; negative: the point stage computes it itself)
    const float es_gate = (PINN_GATE_FAST && SPEC != 0 && SH::has_ic(A)) ? expf(-params_[A.off_ls]) : -1.0f;
    constexpr int SKEW = (TEAMS2 && !SPLIT) ? PINN_TEAM_SKEW : 0;
    constexpr bool NOTOPB = PINN_NOTOPB != 0;
    if (SKEW > 0 && team == 1)
        for (int i = 0; i < SKEW; ++i) tsync();
    // (two teams: both run as many rounds as team 0 has tiles -- a team without a tile in the last round works on an empty
    //  one: zero points, every sample invalid, contributions zero -- so that the barriers match)
    // (virtual workgroups: no hardware barrier in here, so each runs over its own tiles only)
    for (long long tile0 = A.tile_begin + (VWG ? (long long)vbid : (long long)PINN_BID * TEAMS); tile0 < ntiles; tile0 += vnblk, tile_parity ^= 1) {
        const long long tile = VWG ? tile0 : tile0 + team;
        const long long base = tile * T;
        if (WGX && train) {
            // (debug flag 4, timing experiments only: every tile writes the first tile's slab -- stores stay in L2)
            const size_t tl = PINN_DBG(A, 4) ? 0 : (size_t)(tile - A.tile_begin);
            slab = A.slab + tl * C::slab_vec4_per_wg(lh, SKIPS ? 2 * A.n_skips : 0);
            gzs = A.gzslab + tl * C::gz_vec4_per_tile(lh);
        }
        float* xs_t = xs_base + tile_parity * T * PINN_XS_LD;
        float* xs_next = xs_base + (tile_parity ^ 1) * T * PINN_XS_LD;
#ifndef PINN_PTALL
#define PINN_PTALL 1            // 1: the Dirichlet-box kernels with 16-point tiles (product), 2: every shape-specialised kernel (experiment), 0: off
#endif
        // PTALL: every lane runs the point stage of ITS point(s) -- lane (lr, any lq, any wave) owns points mt * 16 + lr -- so the
        // upstream gradient gnet is in the registers of every lane that needs it and the tile has one barrier and one LDS round trip
        // less. Same-box A/B (round 5, profiles/r05_headline_ab.txt): -1.1 % on BASELINE config 2 (16-point tiles, Dirichlet box: the
        // point stage is ~40 instructions), +6.8 % on config 4 (32-point tiles: two points per lane and the IC gate's exponentials in
        // every wave) -- so only where the stage is small: PinnShape 1, one row tile, at most four waves per team
        constexpr bool PTALL = !PROG && ((PINN_PTALL == 2 && SPEC != 0) || (PINN_PTALL == 1 && SPEC == 1 && MT == 1 && NW <= 4));
        PinnPointPre<ND, N2> ppre, ppre_all[PTALL ? MT : 1];
        if constexpr (PTALL) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                pinn_point_prefetch<ND, N2, SPECP>(A, params_, base + mt * 16 + lr, base + mt * 16 + lr < A.n_points, pregs, T, ppre_all[mt], aux_);
        } else {
            if (pt_thread) pinn_point_prefetch<ND, N2, SPECP>(A, params_, base + ptid, base + ptid < A.n_points, pregs + ptid, T, ppre, aux_);
        }
        PH(0)

        float* cur = bufA;
        float* nxt = bufB;
        f32x4 svtop[NTW][MT][S];       // saved jets of the LAST hidden activation: they never leave the registers
        f32x4 htop[NTW][MT][S];        // ... and its activations (last-layer dot and dWL need them again)
        // whole-layer weight fragments are fetched one phase ahead (L2 latency hidden behind the previous epilogue)
        // (not in the VAR 8 kernels: with the four-way activation code between the prefetch and its first use, hipcc
        //  7.2 produced a width-64 kernel whose last prefetched K quad arrived wrong on gfx950 -- reproducible, cured by
        //  -amdgpu-waitcnt-forcezero, by dropping the prefetch, or by the two-way activation)
        constexpr bool WPF = (HP <= 64) && !(VAR & 4) && !(VAR & 8) && !SPLIT;
        constexpr int NQ = HP / 16;
        f32x4 wall[WPF ? NQ : 1][NTW];
        f32x4 biasn[NTW];              // bias of the NEXT hidden layer, fetched with its weights
        auto load_wall = [&](const float* Wl) {
#pragma unroll
            for (int j = 0; j < NTW; ++j) biasn[j] = pinn_ld4(Wl + HP * HP + unit0(j));
#pragma unroll
            for (int q = 0; q < (WPF ? NQ : 1); ++q)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
                    wall[q][j] = pinn_ld4(Wl + ((wave * NTW + j) * 16 + lr) * HP + 16 * q + 4 * lq);
        };
        if (WPF && lh > 0) load_wall(params_ + A.off_wh);
        pinn_s16x8 wsf[SPK][3];       // split-bf16: forward weight fragments of the NEXT hidden layer, one phase ahead
        if constexpr (SPLIT) {
            sp_weights(0, 0, wsf);
#pragma unroll
            for (int j = 0; j < NTW; ++j) biasn[j] = pinn_ld4(params_ + A.off_wh + HP * HP + unit0(j));
        }
        f32x4 hskip[SKIPS ? NTW : 1][SKIPS ? MT : 1][S];      // activations carried by the open skip connection
        const PinnAct act0 = act_at(0);
        const bool src_pre0 = SRCPRE && skip_from(0) >= 0 && ((A.skip_src_pre >> skip_from(0)) & 1);
        // ---- (1) first layer on the VALU: z0 = W1 x + b1, z_k = W1[:, col_k], z_kk = 0 ------------------------
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n0 = unit0(j);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int pt = mt * 16 + lr;
                f32x4 hv[S], sv[S], z0v;
                // point row and the four weight rows as b128 reads (all issued together: one LDS latency)
                const f32x4 xlo = pinn_ld4(xs_t + pt * PINN_XS_LD), xhi = pinn_ld4(xs_t + pt * PINN_XS_LD + 4);
                const f32x4 b1v = pinn_ld4(b1s + n0);
                f32x4 wlo[4], whi[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    wlo[r] = pinn_ld4(W1s + (n0 + r) * PINN_XS_LD);
                    whi[r] = pinn_ld4(W1s + (n0 + r) * PINN_XS_LD + 4);      // columns >= d are zero in both operands
                }
#if defined(PINN_DUMP_NET) && !defined(PINN_EMU)
                // what do the weight registers hold when the arithmetic starts? column 0 of the four rows, copied by VALU moves
                // right behind an explicit lgkmcnt(0)
                f32x4 snap;
                {
                    PINN_SCHED_BARRIER();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    float s0_, s1_, s2_, s3_;
                    asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                                 : "=&v"(s0_), "=&v"(s1_), "=&v"(s2_), "=&v"(s3_) : "v"(wlo[0][0]), "v"(wlo[1][0]), "v"(wlo[2][0]), "v"(wlo[3][0]));
                    snap = f32x4{s0_, s1_, s2_, s3_};
                    PINN_SCHED_BARRIER();
                }
#endif
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + r;
                    float z[S], h[S];
                    float z0 = b1v[r];
#pragma unroll
                    for (int c = 0; c < 4; ++c) z0 = fmaf(wlo[r][c], xlo[c], z0);
                    if (d > 4) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) z0 = fmaf(whi[r][c], xhi[c], z0);
                    }
                    z[0] = z0;
#pragma unroll
                    for (int k = 0; k < ND; ++k) z[1 + k] = pinn_dir_weight<pinn_dir_x(N2)>(W1s + n * PINN_XS_LD, SH::dir(A, k));
#pragma unroll
                    for (int s = 1 + ND; s < S; ++s) z[s] = 0.0f;              // z_kk = z_kkk = 0 in the first layer
                    pinn_jet_fwd<ND, N2, COMB>(z, with_poly(act0), h, cw);
#pragma unroll
                    for (int s = 0; s < S; ++s) { hv[s][r] = h[s]; sv[s][r] = (s == 0) ? pinn_act_saved(h[0], z[0], act0) : z[s]; }
                    if (SRCPRE) z0v[r] = z[0];
#if defined(PINN_DUMP_NET) && !defined(PINN_EMU)
                    snap[r] = z[0];             // (the dump's second per-lane row: the pre-activation the activation was evaluated on)
#endif
                }
#if defined(PINN_DUMP_NET) && !defined(PINN_EMU)
                if (j == 0 && mt == 0) {
                    // the inputs of the first layer as this wave saw them: point row, bias, weight rows; then the derivative streams
                    dump_layer(tile, 4, xlo); dump_layer(tile, 5, b1v);
                    dump_layer(tile, 6, wlo[0] + wlo[1] * 3.0f + wlo[2] * 5.0f + wlo[3] * 7.0f);
                    dump_layer(tile, 7, sv[0]);
                    for (int s = 1; s < S && s < 5; ++s) dump_layer(tile, 7 + s, hv[s]);
                    // per lane: the value stream and the first derivative stream of the first layer (behind the checksums: [tile][wave][2][64] x 4)
                    if (A.prof) {
                        f32x4* per_lane = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(A.prof) + 8LL * A.n_points + 64LL * (A.n_points / T + 1));
                        per_lane[((tile * 4 + wave) * 2 + 0) * 64 + lane] = hv[0];
                        per_lane[((tile * 4 + wave) * 2 + 1) * 64 + lane] = snap;
                    }
                }
#endif
                if (SKIPS && skip_from(0) >= 0) {
                    // ('f R a': the skip carries the z-jets, not act(z): z_0 kept aside, the derivative jets are what sv holds)
                    const bool park = NEST && ((A.skip_outer >> skip_from(0)) & 1);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const f32x4 carried = (SRCPRE && src_pre0) ? (s == 0 ? z0v : sv[s]) : hv[s];
                        if (NEST && park) skip_park(skip_from(0), s, j, mt, carried);
                        else hskip[SKIPS ? j : 0][SKIPS ? mt : 0][s] = carried;
                    }
                }
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if constexpr (SPLIT) sp_store(cur, s * T + pt, n0, hv[s]);
                    else pinn_st4(cur + (s * T + pt) * LDA + n0, hv[s]);
                }
#if defined(PINN_DUMP_NET) && !defined(PINN_EMU)
                if (j == 0 && mt == 0) dump_layer(tile, 0, hv[0]);
#endif
                if (lh == 0) {
#pragma unroll
                    for (int s = 0; s < S; ++s) htop[j][mt][s] = hv[s];
                }
                if (train) {
                    if (lh == 0) {
#pragma unroll
                        for (int s = 0; s < S; ++s) svtop[j][mt][s] = sv[s];
                    } else {
                        if (WGX) pinn_st4_stream<SNT>(slab_at(0, 0, j, mt), sv[0]);
                        else if (!(PINN_ABL & 2)) *slab_at(0, 0, j, mt) = sv[0];     // z_k = W1[:, col_k] and z_kk = 0 are rebuilt in the reverse half
                    }
                }
            }
        }
        PH(1)
        tsync();
        PH(2)

        // ---- (2) hidden layers: Z^T = W H^T (MFMA: A = weight fragment, B = activations), jets on accumulators --------
        for (int li = 0; li < lh; ++li) {
            const float* Wl = params_ + A.off_wh + (size_t)li * A.hidden_stride;
            const float* bl = Wl + HP * HP;
            const PinnAct act = act_at(li + 1);
            const int sk_in = skip_into(li + 1), sk_out = skip_from(li + 1);
            f32x4 acc[NTW][MT][S];
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int s = 0; s < S; ++s) acc[j][mt][s] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) {
                if constexpr (SPW) {
                    sp_gemm_wide(cur, li, 0, acc);
                } else {
                    sp_gemm(cur, wsf, acc);
                }
            } else {
                // software pipeline over the K quads: the operands of quad q+1 are in flight while the S*MT*NTW*4 MFMAs
                // of quad q issue, accumulators interleaved (an accumulator is re-used every S*MT*NTW issues, far
                // beyond the 40-cycle dependent latency). sched_barrier pins that order (the register-pressured
                // scheduler otherwise sinks every load next to its first use).
                // (WA: quads the weight fragments run ahead. Two or three for the weights that come from global memory / L2 -- widths
                //  >= 128, 20 MFMAs per quad are about one L2 round trip -- measured +-0 / +1.4 %: DESIGN.md section 6a)
                constexpr int WA = 1;
                constexpr int NQF = HP / 16;
                f32x4 wf[WA + 1][NTW], hf[2][MT][S];
                auto load_w = [&](int q, f32x4 (&w)[NTW]) {
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        w[j] = WPF ? wall[WPF ? q : 0][j]
                                   : pinn_ld4(Wl + ((wave * NTW + j) * 16 + lr) * HP + 16 * q + 4 * lq);
                };
                auto load_h = [&](int q, f32x4 (&h)[MT][S]) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int s = 0; s < S; ++s)
                            h[mt][s] = pinn_ld4(cur + (s * T + mt * 16 + lr) * LDA + 16 * q + 4 * lq);
                };
#pragma unroll
                for (int i = 0; i < WA; ++i)
                    if (i < NQF) load_w(i, wf[i]);
                load_h(0, hf[0]);
#pragma unroll
                for (int q = 0; q < NQF; ++q) {
                    PINN_SCHED_BARRIER();
                    if (q + WA < NQF) load_w(q + WA, wf[(q + WA) % (WA + 1)]);
                    if (q + 1 < NQF) load_h(q + 1, hf[(q + 1) & 1]);
                    if (!PINN_SCHED_IL) PINN_SCHED_BARRIER();
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int s = 0; s < S; ++s)
#pragma unroll
                                for (int j = 0; j < NTW; ++j)
                                    acc[j][mt][s] = pinn_mfma16(wf[q % (WA + 1)][j][m], hf[q & 1][mt][s][m], acc[j][mt][s]);
                    if (q + 1 < NQF) pinn_sched_interleave<4 * MT * S * NTW, MT * S + (WPF ? 0 : NTW)>();
                    PINN_SCHED_BARRIER();
                }
            }
            PH(3)
            if (ONEBUF) tsync();                       // in place: every wave must be done reading h_{l-1}
            f32x4 biasv[NTW];
#pragma unroll
            for (int j = 0; j < NTW; ++j) biasv[j] = (WPF || SPLIT) ? biasn[j] : pinn_ld4(bl + unit0(j));
            PINN_SCHED_BARRIER();
            if (WPF && li + 1 < lh) load_wall(Wl + A.hidden_stride);
            if constexpr (SPLIT) {
                if (li + 1 < lh) {
                    sp_weights(li + 1, 0, wsf);
#pragma unroll
                    for (int j = 0; j < NTW; ++j) biasn[j] = pinn_ld4(Wl + A.hidden_stride + HP * HP + unit0(j));
                }
            }
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int n0 = unit0(j);
                const f32x4 bias = biasv[j];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int pt = mt * 16 + lr;
                    f32x4 hv[S], sv[S], z0v;
                    // '+' in front of the activation: the jets of the skipped activations join the pre-activation jets
                    const bool pre_in = SKIPS && sk_in >= 0 && ((A.skip_pre >> sk_in) & 1);
                    const bool src_pre = SRCPRE && sk_out >= 0 && ((A.skip_src_pre >> sk_out) & 1);
                    // the jets that join here: out of the register set, or (outer skip of a nest) back from the skip's slab slot
                    const bool in_parked = NEST && sk_in >= 0 && ((A.skip_outer >> sk_in) & 1);
                    f32x4 hin[SKIPS ? S : 1];
                    if (SKIPS && sk_in >= 0) {
#pragma unroll
                        for (int s = 0; s < S; ++s)
                            hin[SKIPS ? s : 0] = (NEST && in_parked) ? *slab_at(lh + 1 + sk_in, s, j, mt) : hskip[SKIPS ? j : 0][SKIPS ? mt : 0][s];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float z[S], h[S];
#pragma unroll
                        for (int s = 0; s < S; ++s) z[s] = acc[j][mt][s][r];
                        z[0] += bias[r];
                        if (SKIPS && pre_in) {
#pragma unroll
                            for (int s = 0; s < S; ++s) z[s] += hin[SKIPS ? s : 0][r];
                        }
                        pinn_jet_fwd<ND, N2, COMB>(z, with_poly(act), h, cw);
#pragma unroll
                        for (int s = 0; s < S; ++s) { hv[s][r] = h[s]; sv[s][r] = (s == 0) ? pinn_act_saved(h[0], z[0], act) : z[s]; }
                        if (SRCPRE) z0v[r] = z[0];
                    }
                    if (SKIPS && sk_in >= 0 && !pre_in) {
                        // '+' behind the activation: add the activations saved at 'R'; the reverse half needs them again (slab slot of the skip)
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            const f32x4 hs = hin[SKIPS ? s : 0];
                            hv[s] += hs;
                            if (train && !(NEST && in_parked)) {            // (a parked skip's jets are in the slot already)
                                if (WGX) pinn_st4_stream<SNT>(slab_at(lh + 1 + sk_in, s, j, mt), hs);
                                else *slab_at(lh + 1 + sk_in, s, j, mt) = hs;
                            }
                        }
                    }
                    if (SKIPS && sk_out >= 0) {
                        // ('f R a': the skip carries the z-jets, not act(z): z_0 kept aside, the derivative jets are what sv holds)
                        const bool park = NEST && ((A.skip_outer >> sk_out) & 1);
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            const f32x4 carried = (SRCPRE && src_pre) ? (s == 0 ? z0v : sv[s]) : hv[s];
                            if (NEST && park) skip_park(sk_out, s, j, mt, carried);
                            else hskip[SKIPS ? j : 0][SKIPS ? mt : 0][s] = carried;
                        }
                    }
#if defined(PINN_DUMP_NET) && !defined(PINN_EMU)
                    if (j == 0 && mt == 0) dump_layer(tile, li + 1, hv[0]);
#endif
                    if (li + 1 == lh) {
#pragma unroll
                        for (int s = 0; s < S; ++s) htop[j][mt][s] = hv[s];
                    } else {
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            if constexpr (SPLIT) sp_store(nxt, s * T + pt, n0, hv[s]);
                            else pinn_st4(nxt + (s * T + pt) * LDA + n0, hv[s]);
                        }
                    }
                    if (train) {
                        if (li + 1 == lh) {
#pragma unroll
                            for (int s = 0; s < S; ++s) svtop[j][mt][s] = sv[s];
                        } else {
#pragma unroll
                            for (int s = 0; s < S; ++s) {
                                if (WGX) pinn_st4_stream<SNT>(slab_at(li + 1, s, j, mt), sv[s]);
                                else if (!(PINN_ABL & 2)) *slab_at(li + 1, s, j, mt) = sv[s];
                            }
                        }
                    }
                }
            }
            PH(4)
            if (!(NOTOPB && li + 1 == lh)) tsync();
            PH(5)
            float* tmp = cur; cur = nxt; nxt = tmp;
        }

        // ---- (3) last layer (out = 1): net_s[pt] = WL . h_s[pt]; per-wave partials, summed by the point stage ------
        {
            // (weights read once, all dot products first, then the row sums, then ONE predicated block of stores: written
            //  the short way the compiler re-read WL and opened an exec-mask region for every (mt, s) pair, eight LDS
            //  round trips in a row)
            f32x4 wlv[NTW];
#pragma unroll
            for (int j = 0; j < NTW; ++j) wlv[j] = pinn_ld4(WLs + unit0(j));
            float part[MT][S];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    float acc1 = 0.0f;
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        const f32x4 hv = htop[j][mt][s];
                        acc1 = fmaf(hv[0], wlv[j][0], acc1); acc1 = fmaf(hv[1], wlv[j][1], acc1);
                        acc1 = fmaf(hv[2], wlv[j][2], acc1); acc1 = fmaf(hv[3], wlv[j][3], acc1);
                    }
                    part[mt][s] = acc1;
                }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int s = 0; s < S; ++s) part[mt][s] = pinn_rows_sum(part[mt][s]);
            if (lq == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int s = 0; s < S; ++s) netp[(wave * S + s) * T + mt * 16 + lr] = part[mt][s];
            }
        }
        PH(6)
        tsync();
        PH(7)

        // ---- (4) ansatz + residual + their reverse, one thread per point; all threads: stage the NEXT tile's points ------
        store_points(xs_next);
        fetch_points(tile + 2LL * vnblk);
        float gnet_r[PTALL ? MT : 1][S];
        if constexpr (PTALL) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int pt = mt * 16 + lr;
                float net[S];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    float v = (s == 0) ? bL : 0.0f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) v += netp[(w * S + s) * T + pt];
                    net[s] = v;
                }
                PinnPointOut<ND, N2> po;
                pinn_point_stage<ND, N2, false, COMB, SPECP>(A, params_, net, xs_t + pt * PINN_XS_LD, base + pt, base + pt < A.n_points,
                                                            pregs, padj, T, ppre_all[mt], po, es_gate);
#pragma unroll
                for (int s = 0; s < S; ++s) gnet_r[PTALL ? mt : 0][s] = po.gnet[s];
                if (wave == ptw && lq == 0) { sum_loss += po.loss; sum_ls += po.g_ls; sum_bl += po.gnet[0]; }     // (one lane per point counts)
            }
        } else
        if (pt_thread) {
            const int pt = ptid;
            float net[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float v = (s == 0) ? bL : 0.0f;
#pragma unroll
                for (int w = 0; w < NW; ++w) v += netp[(w * S + s) * T + pt];
                net[s] = v;
            }
#if defined(PINN_DUMP_NET) && !defined(PINN_EMU)
            // experiment builds (tools/var2.sh): the network's output streams of every point, [S][N] floats behind A.prof, so that two
            // runs can be compared point by point (tools/diff_runs.py --net)
            if (A.prof && base + pt < A.n_points) {
#pragma unroll
                for (int s = 0; s < S; ++s) reinterpret_cast<float*>(A.prof)[(long long)s * A.n_points + base + pt] = net[s];
                // ... and the coordinates the tile worked on (rows S, S + 1 of the dump)
                for (int c = 0; c < 2 && S + c < 8; ++c)
                    reinterpret_cast<float*>(A.prof)[(long long)(S + c) * A.n_points + base + pt] = xs_t[pt * PINN_XS_LD + c];
            }
#endif
            PinnPointOut<ND, N2> po;
            pinn_point_stage<ND, N2, SPEC == 0 || PROG, COMB, SPECP>(A, params_, net, xs_t + pt * PINN_XS_LD, base + pt, base + pt < A.n_points,
                                     pregs + pt, padj + pt, T, ppre, po, es_gate);
#pragma unroll
            for (int s = 0; s < S; ++s) gnetb[s * T + pt] = po.gnet[s];
            sum_loss += po.loss; sum_ls += po.g_ls; sum_bl += po.gnet[0];
            if (SPEC == 0) sum_ic += po.g_ic;
        }
        PH(8)
        if (!PTALL || lh == 0) tsync();         // (PTALL: nothing crossed the LDS, the next write to `netp` is a whole tile of barriers away; a net without
                                                //  hidden->hidden layers has no other barrier between the staging of the next tile's points and their first reader)
        PH(9)
        if (!train) continue;

        // ---- (5) reverse through the last layer: gh_s = gnet_s * WL ; dWL += sum gnet_s h_s -------------------------
        f32x4 g[NTW][MT][S];
        f32x4 sv[NTW][MT][S];
        // saved jets of the activation BELOW the one being reversed: fetched one phase ahead (L2 latency behind a GEMM phase)
        f32x4 svn[NTW][MT][S];
        auto load_saved = [&](int a, f32x4 (&dst)[NTW][MT][S]) {
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (a == 0) {
                        dst[j][mt][0] = (PINN_ABL & 4) ? f32x4{0.1f, 0.2f, 0.3f, 0.4f} : *slab_at(0, 0, j, mt);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
#pragma unroll
                            for (int k = 0; k < ND; ++k)
                                dst[j][mt][1 + k][r] = pinn_dir_weight<pinn_dir_x(N2)>(W1s + (unit0(j) + r) * PINN_XS_LD, SH::dir(A, k));
#pragma unroll
                            for (int s = 1 + ND; s < S; ++s) dst[j][mt][s][r] = 0.0f;
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < S; ++s) dst[j][mt][s] = (PINN_ABL & 4) ? f32x4{0.1f, 0.2f, 0.3f, 0.4f} * (float)(s + 1) : *slab_at(a, s, j, mt);
                    }
                }
        };
        // (only where S*MT*NTW jets leave register head-room: measured -3 % on cfg2/cfg5, +1..5 % on cfg3/cfg4)
#ifndef PINN_SVPF_MAX
#define PINN_SVPF_MAX 8
#endif
#ifndef PINN_SVPF_WIDE_MAX
#define PINN_SVPF_WIDE_MAX 4       // 8-wave kernels: registers first -- only the shape-specialised ones with S <= 4 streams at width 128 have
#endif                             // room (same-box A/B, round 4: skip128 tile kernel -1.2 %; the S = 5 kernel of config 3 spills and loses 4 %)
        constexpr bool SVPF = (S * MT * NTW <= ((NW <= 4) ? PINN_SVPF_MAX : ((VAR & 48) ? PINN_SVPF_WIDE_MAX : 0))) && !ONEBUF;
        if (SVPF && lh > 0) load_saved(lh - 1, svn);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n0 = unit0(j);
            const f32x4 wl = pinn_ld4(WLs + n0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int pt = mt * 16 + lr;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    sv[j][mt][s] = svtop[j][mt][s];
                    const float gn = PTALL ? gnet_r[PTALL ? mt : 0][s] : gnetb[s * T + pt];
                    g[j][mt][s] = wl * gn;
                    accWL[j] += htop[j][mt][s] * gn;
                }
            }
        }

        // ---- (6) reverse through the activations / hidden layers; the layer index is unrolled so that the dW
        //          accumulators are addressed statically (they must stay in registers) ----------------------------
        auto act_reverse = [&](int a, f32x4 (&gz)[NTW][MT][S], f32x4 (&bacc)[NTW]) {
            // gz_a = jet-reverse(gh, saved_a);  db_a += sum_pt gz_a,0 (DPP row sum over the 16 points of the lane row)
            const PinnAct act = act_at(a);
            if (SKIPS) {
                // h_out(a) feeds a later '+': its gradient arrives through the skip slot; h_out(a) = act(z_a) + skipped
                // activations: the whole gradient is handed down the skip (slot re-used: its activations are consumed)
                // (a '+' in front of the activation hands down gz_a instead, below)
                const int k_out_any = skip_from(a), k_in = skip_into(a);
                const int k_out = (k_out_any >= 0 && !(SRCPRE && ((A.skip_src_pre >> k_out_any) & 1))) ? k_out_any : -1;    // (a skip that left in front of
                                                                                                              //  the activation returns to gz, below)
                const bool post_in = k_in >= 0 && !((A.skip_pre >> k_in) & 1);
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            if (k_out >= 0) g[j][mt][s] += *slab_at(skip_grad_slot(k_out), s, j, mt);
                            if (post_in) *slab_at(skip_grad_slot(k_in), s, j, mt) = g[j][mt][s];
                        }
            }
            // z_a fed a later '+' itself ('f R a'): the gradient that comes back along that skip belongs to gz_a
            int src_pre_k = -1;
            if (SRCPRE) {
                const int k = skip_from(a);
                if (k >= 0 && ((A.skip_src_pre >> k) & 1)) src_pre_k = k;
            }
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
                // (PINN_BIAS_EARLY_LD: the row this wave adds its bias-gradient sums to, read by every lane -- a broadcast -- in front of the
                //  jet arithmetic; only this wave writes these units, and LDS executes a wave's accesses in order)
                f32x4 bold = f32x4{0.f, 0.f, 0.f, 0.f};
                if (PINN_BIAS_EARLY_LD && !REGB) bold = pinn_ld4(accB + a * HP + unit0(j));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float gh1[S], sv1[S], gz1[S];
#pragma unroll
                        for (int s = 0; s < S; ++s) { gh1[s] = g[j][mt][s][r]; sv1[s] = sv[j][mt][s][r]; }
                        pinn_jet_bwd<ND, N2, COMB>(gh1, sv1, act, gz1, cw);
#pragma unroll
                        for (int s = 0; s < S; ++s) gz[j][mt][s][r] = gz1[s];
                        bsum[r] += gz1[0];
                    }
                if (SRCPRE && src_pre_k >= 0) {             // (outside the loop over r: a branch in there splits the jet arithmetic)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            const f32x4 back = *slab_at(skip_grad_slot(src_pre_k), s, j, mt);
                            gz[j][mt][s] += back;
                            if (s == 0) bsum += back;
                        }
                }
                if (REGB) {
                    bacc[j] += bsum;
                } else {
                    bsum = pinn_row_sum16_v4(bsum);
                    if (lr == 0) {
                        float* dst = accB + a * HP + unit0(j);
                        pinn_st4(dst, (PINN_BIAS_EARLY_LD ? bold : pinn_ld4(dst)) + bsum);
                    }
                }
            }
            if (SKIPS) {
                const int k_in = skip_into(a);
                if (k_in >= 0 && ((A.skip_pre >> k_in) & 1)) {
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int s = 0; s < S; ++s) *slab_at(skip_grad_slot(k_in), s, j, mt) = gz[j][mt][s];
                }
            }
        };
        // after the forward half `nxt` still holds h_{lh-1} (the input of the last hidden layer): the top reverse step
        // uses it in place and stages only gz
        auto hidden_reverse = [&](int a, f32x4 (&dw)[DWG ? 1 : NT][NTW], f32x4 (&bacc)[NTW]) {
            f32x4 gz[NTW][MT][S];
            act_reverse(a, gz, bacc);
            const bool top = (a == lh);
            if (top && !ONEBUF) { float* tmp = cur; cur = nxt; nxt = tmp; }   // cur = h_{a-1}, nxt = free (receives gz)
            // recompute h_{a-1} from its saved jets (kept in sv for the next step); stage h_{a-1} and gz_a for the GEMMs
            // (WGX: the weight gradient is pinn_wgrad_kernel's job -- nothing of h_{a-1} is needed here, gz_a goes to LDS for
            //  the data-gradient GEMM and to HBM for that kernel)
            f32x4 hv[NTW][MT][S];               // (WGX: never touched)
            if (!SVPF) load_saved(a - 1, svn);
            const PinnAct act = act_at(a - 1);
            int sk_prev = skip_into(a - 1);
            if (SKIPS && sk_prev >= 0 && ((A.skip_pre >> sk_prev) & 1)) sk_prev = -1;      // (joined in front of the activation: h_out = act(z))
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int s = 0; s < S; ++s) sv[j][mt][s] = svn[j][mt][s];     // fetched one phase ago (or just now)
                    if constexpr (!WGX) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float sv1[S], h[S];
#pragma unroll
                            for (int s = 0; s < S; ++s) sv1[s] = sv[j][mt][s][r];
                            pinn_jet_recompute<ND, N2, COMB>(sv1, act, h, cw);
#pragma unroll
                            for (int s = 0; s < S; ++s) hv[j][mt][s][r] = h[s];
                        }
                        if (SKIPS && sk_prev >= 0) {
#pragma unroll
                            for (int s = 0; s < S; ++s) hv[j][mt][s] += *slab_at(lh + 1 + sk_prev, s, j, mt);
                        }
                    }
                }
            }
            auto stage = [&](float* buf, f32x4 (&v)[NTW][MT][S]) {
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            if constexpr (SPLIT) sp_store(buf, s * T + mt * 16 + lr, unit0(j), v[j][mt][s]);
                            else pinn_st4(buf + (s * T + mt * 16 + lr) * LDA + unit0(j), v[j][mt][s]);
                        }
            };
            // B fragments of the weight-gradient GEMM (h_{a-1}): lane (lr, lq) needs h[pt = wg_pt(m)][its unit column]
            float hfrag[(ONEBUF && !WGX) ? MT * S : 1][NTW][4];
            // gz_a goes to HBM behind the data-gradient GEMM instead of in front of it where the registers allow (width 128:
            // -1.5 % in a same-box A/B; the GEMM's first weight loads no longer queue behind 20 stores per lane)
#ifndef PINN_GZ_LATE_MAX_HP
#define PINN_GZ_LATE_MAX_HP 128
#endif
            constexpr bool GZ_LATE = HP <= PINN_GZ_LATE_MAX_HP;
            auto store_gz = [&]() {
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int s = 0; s < S; ++s) pinn_st4_stream<SNT>(gz_at(a, s, j, mt), gz[j][mt][s]);
            };
            if constexpr (WGX) {
                stage(nxt, gz);
                if (!GZ_LATE) store_gz();
            } else if constexpr (ONEBUF) {
                // one LDS buffer: h_{a-1} -> LDS -> fragments in registers, then gz_a takes its place
                if (!top) { stage(cur, hv); tsync(); }
#pragma unroll
                for (int ms = 0; ms < MT * S; ++ms)
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            hfrag[ONEBUF ? ms : 0][j][m] =
                                cur[((ms % S) * T + (ms / S) * 16 + wg_pt(m)) * LDA + (wave * NTW + j) * 16 + lr];
                tsync();
                stage(nxt, gz);
            } else {
                if (!top) stage(cur, hv);
                stage(nxt, gz);
            }
            PH(10)
            tsync();
            PH(11)
            if (SVPF && a >= 2) load_saved(a - 2, svn);    // in flight during the two GEMMs below
            const int li = a - 1;
            pinn_s16x8 wsb[SPK][3];       // split-bf16: W^T fragments of this layer (L2 round trip behind the weight-gradient GEMM)
            if constexpr (SPLIT) sp_weights(li, 1, wsb);
            const float* Wl = params_ + A.off_wh + (size_t)li * A.hidden_stride;
            float wqall[WPF ? NQ : 1][NTW][4];
            if (WPF && !WTL) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            wqall[WPF ? q : 0][j][m] = Wl[(16 * q + 4 * lq + m) * HP + (wave * NTW + j) * 16 + lr];
            }
            // weight gradient: dW_li[out][in] += sum_{s,pt} gz_s[pt][out] * h_s[pt][in]   (A = gz^T, B = h)
            if constexpr (WGX) {
                // pinn_wgrad_kernel
            } else if constexpr (SPLIT) {
                // K = the S * T (stream, point) pairs of the tile in blocks of 32; k slot (lq, e) of block kb:
                //   T = 16: stream 2 kb + (lq >> 1), point 4 (lq & 1) + e (e < 4) / 8 + 4 (lq & 1) + e - 4
                //   T = 32: stream kb,               point 4 lq + e (e < 4)       / 16 + 4 lq + e - 4
                // (any bijection works as long as both operands use it; this one keeps the transpose reads conflict-free).
                // ds_read_b64_tr_b16: lane i of a 16-lane group addresses (row p0 + i / 4, units 16 o + 4 (i % 4) ..) and receives
                // the four points of unit 16 o + i -- two reads per plane make one 8-slot fragment.
                const char* gb = reinterpret_cast<const char*>(nxt);
                const char* hb = reinterpret_cast<const char*>(cur);
                auto tr_frag = [&](const char* base, int row0, int row1, int ucol, pinn_s16x8 (&f)[3]) {
                    // ucol = first unit of the 16-unit column block; this lane's 8 bytes: units ucol + 4 (lr & 3) .. + 3
                    const int r0 = row0 + (lr >> 2), r1 = row1 + (lr >> 2), u = ucol + 4 * (lr & 3);
                    const int o0 = pinn_sp_off<C::SP_ROW_BYTES>(r0, u >> 3) + ((u >> 2) & 1) * 8;
                    const int o1 = pinn_sp_off<C::SP_ROW_BYTES>(r1, u >> 3) + ((u >> 2) & 1) * 8;
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const pinn_s16x4 x = pinn_lds_tr16(base + p * C::SP_PLANE_BYTES + o0);
                        const pinn_s16x4 y = pinn_lds_tr16(base + p * C::SP_PLANE_BYTES + o1);
                        f[p] = pinn_s16x8{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
                    }
                };
                constexpr int KBW = (S * T) / 32, OG = PINN_SP_OG, NOG = NT / OG, STEPS = KBW * NOG, NB = PINN_SP_PIPE_W ? 2 : 1;
                constexpr int pa[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, pb[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};     // (a part, b part), small products first
                auto rows_of = [&](int kb, int& r0, int& r1) {
                    const int srow = (T == 16) ? (2 * kb + (lq >> 1)) * T : kb * T;
                    r0 = srow + ((T == 16) ? 4 * (lq & 1) : 4 * lq);
                    r1 = srow + ((T == 16) ? 8 + 4 * (lq & 1) : 16 + 4 * lq);
                };
                // B fragments (h_{a-1}, this wave's 16 input units) of every K block up front; the A fragments (gz_a, 16 output
                // units per tile row o) two tile rows per step, one step ahead of their twelve MFMAs
                pinn_s16x8 bfr[PINN_SP_PIPE_W ? KBW : 1][3], afr[NB][OG][3];
                if (PINN_SP_PIPE_W) {
#pragma unroll
                    for (int kb = 0; kb < KBW; ++kb) {
                        int r0, r1;
                        rows_of(kb, r0, r1);
                        tr_frag(hb, r0, r1, wave * 16, bfr[PINN_SP_PIPE_W ? kb : 0]);
                    }
                }
                auto load_a = [&](int step, pinn_s16x8 (&f)[OG][3]) {
                    int r0, r1;
                    rows_of(step / NOG, r0, r1);
#pragma unroll
                    for (int i = 0; i < OG; ++i) tr_frag(gb, r0, r1, ((step % NOG) * OG + i) * 16, f[i]);
                };
                if (PINN_SP_PIPE_W) load_a(0, afr[0]);
#pragma unroll
                for (int step = 0; step < STEPS; ++step) {
                    PINN_SCHED_BARRIER();
                    const int kb = step / NOG, o0 = (step % NOG) * OG;
                    if (PINN_SP_PIPE_W) {
                        if (step + 1 < STEPS) load_a(step + 1, afr[(step + 1) % NB]);
                    } else {
                        if (step % NOG == 0) {
                            int r0, r1;
                            rows_of(kb, r0, r1);
                            tr_frag(hb, r0, r1, wave * 16, bfr[0]);
                        }
                        load_a(step, afr[0]);
                    }
                    if (!PINN_SCHED_IL) PINN_SCHED_BARRIER();
#pragma unroll
                    for (int t = 9 - PINN_SP_NPROD; t < 9; ++t)
#pragma unroll
                        for (int i = 0; i < OG; ++i)
                            dw[o0 + i][0] = pinn_mfma16_bf16(afr[step % NB][i][pa[t]], bfr[PINN_SP_PIPE_W ? kb : 0][pb[t]], dw[o0 + i][0]);
                    if (PINN_SP_PIPE_W && step + 1 < STEPS) pinn_sched_interleave<6 * OG, 6 * OG>();
                    PINN_SCHED_BARRIER();
                }
            } else if constexpr (!DWG) {
                // register accumulators, software pipeline over the (mt, s) row tiles, NT*NTW accumulators interleaved
                float bq[2][NTW][4], aq[2][NT][4];
                auto load_ms = [&](int ms, float (&b)[NTW][4], float (&a_)[NT][4]) {
                    const int mt = ms / S, s = ms % S;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int row = (s * T + mt * 16 + wg_pt(m)) * LDA;
#pragma unroll
                        for (int j = 0; j < NTW; ++j) b[j][m] = cur[row + (wave * NTW + j) * 16 + lr];
#pragma unroll
                        for (int o = 0; o < NT; ++o) a_[o][m] = nxt[row + o * 16 + lr];
                    }
                };
                load_ms(0, bq[0], aq[0]);
#pragma unroll
                for (int ms = 0; ms < MT * S; ++ms) {
                    PINN_SCHED_BARRIER();
                    if (ms + 1 < MT * S) load_ms(ms + 1, bq[(ms + 1) & 1], aq[(ms + 1) & 1]);
                    if (!PINN_SCHED_IL) PINN_SCHED_BARRIER();
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int o = 0; o < NT; ++o)
#pragma unroll
                            for (int j = 0; j < NTW; ++j)
                                dw[o][j] = pinn_mfma16(aq[ms & 1][o][m], bq[ms & 1][j][m], dw[o][j]);
                    // (the 4 * (NT + NTW) column reads usually pair up into ds_read2_b32)
                    if (ms + 1 < MT * S) pinn_sched_interleave<4 * NT * NTW, 2 * (NT + NTW)>();
                    PINN_SCHED_BARRIER();
                }
            } else {
                if constexpr (ONEBUF) {
                    // width 256: the kernel sits at its 256-VGPR limit (two waves per SIMD) and already spills; the plain
                    // loop (one output tile row at a time, NTW = 2 chains, no extra buffers) measured 4 % faster than the
                    // blocked / pipelined form below
                    for (int o = 0; o < NT; ++o) {
                        f32x4 dwt[NTW];
#pragma unroll
                        for (int j = 0; j < NTW; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) dwt[j][r] = *dwg_ptr(li, o, j, r);
#pragma unroll
                        for (int ms = 0; ms < MT * S; ++ms) {
                            const int mt = ms / S, s = ms % S;
                            float aq[4];
#pragma unroll
                            for (int m = 0; m < 4; ++m) aq[m] = nxt[(s * T + mt * 16 + wg_pt(m)) * LDA + o * 16 + lr];
#pragma unroll
                            for (int m = 0; m < 4; ++m)
#pragma unroll
                                for (int j = 0; j < NTW; ++j) dwt[j] = pinn_mfma16(aq[m], hfrag[ONEBUF ? ms : 0][j][m], dwt[j]);
                        }
#pragma unroll
                        for (int j = 0; j < NTW; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) *dwg_ptr(li, o, j, r) = dwt[j][r];
                    }
                } else {
                    // accumulators in the workgroup's partial buffer (any depth, any width): OB output tile rows at a time --
                    // OB independent MFMA chains per B fragment (a lone chain stalls on its own 8-pass latency), the tiles of
                    // the NEXT block fetched from the partial buffer (L2) while this block computes, the LDS operands of the
                    // next (mt, s) row tile read between the MFMAs of the current one
                constexpr int OB_CHAINS = 4;
                constexpr int OB = (NTW >= OB_CHAINS) ? 1 : ((OB_CHAINS / NTW < NT) ? OB_CHAINS / NTW : NT);   // OB * NTW chains
                    f32x4 dwn[OB][NTW];
                    auto load_dw = [&](int ob, f32x4 (&dst)[OB][NTW]) {
#pragma unroll
                        for (int oo = 0; oo < OB; ++oo)
#pragma unroll
                            for (int j = 0; j < NTW; ++j)
#pragma unroll
                                for (int r = 0; r < 4; ++r) dst[oo][j][r] = *dwg_ptr(li, ob * OB + oo, j, r);
                    };
                    load_dw(0, dwn);
                    for (int ob = 0; ob < NT / OB; ++ob) {
                        f32x4 dwt[OB][NTW];
#pragma unroll
                        for (int oo = 0; oo < OB; ++oo)
#pragma unroll
                            for (int j = 0; j < NTW; ++j) dwt[oo][j] = dwn[oo][j];
                        if (ob + 1 < NT / OB) load_dw(ob + 1, dwn);
                        float aq[2][OB][4], bq[2][NTW][4];
                        auto load_ms = [&](int ms, float (&a_)[OB][4], float (&b)[NTW][4]) {
                            const int mt = ms / S, s = ms % S;
#pragma unroll
                            for (int m = 0; m < 4; ++m) {
                                const int row = (s * T + mt * 16 + wg_pt(m)) * LDA;
#pragma unroll
                                for (int oo = 0; oo < OB; ++oo) a_[oo][m] = nxt[row + (ob * OB + oo) * 16 + lr];
#pragma unroll
                                for (int j = 0; j < NTW; ++j)
                                    b[j][m] = ONEBUF ? hfrag[ONEBUF ? ms : 0][j][m] : cur[row + (wave * NTW + j) * 16 + lr];
                            }
                        };
                        load_ms(0, aq[0], bq[0]);
#pragma unroll
                        for (int ms = 0; ms < MT * S; ++ms) {
                            PINN_SCHED_BARRIER();
                            if (ms + 1 < MT * S) load_ms(ms + 1, aq[(ms + 1) & 1], bq[(ms + 1) & 1]);
                            if (!PINN_SCHED_IL) PINN_SCHED_BARRIER();
#pragma unroll
                            for (int m = 0; m < 4; ++m)
#pragma unroll
                                for (int oo = 0; oo < OB; ++oo)
#pragma unroll
                                    for (int j = 0; j < NTW; ++j)
                                        dwt[oo][j] = pinn_mfma16(aq[ms & 1][oo][m], bq[ms & 1][j][m], dwt[oo][j]);
                            if (ms + 1 < MT * S) pinn_sched_interleave<4 * OB * NTW, 2 * (OB + (ONEBUF ? 0 : NTW))>();
                            PINN_SCHED_BARRIER();
                        }
#pragma unroll
                        for (int oo = 0; oo < OB; ++oo)
#pragma unroll
                            for (int j = 0; j < NTW; ++j)
#pragma unroll
                                for (int r = 0; r < 4; ++r) *dwg_ptr(li, ob * OB + oo, j, r) = dwt[oo][j][r];
                    }
                }
            }
            PH(12)
            // data gradient: GH^T[in][pt] = sum_out W_li[out][in] * gz[pt][out]   (A = W^T fragment, B = gz)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int s = 0; s < S; ++s) g[j][mt][s] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) {
                if constexpr (SPW) sp_gemm_wide(nxt, li, 1, g);
                else sp_gemm(nxt, wsb, g);
            } else {
                constexpr int WAD = 1;                     // (quads the weight fragments run ahead, see the forward GEMM)
                constexpr int NQD = HP / 16;
                float wq[WAD + 1][NTW][4];
                f32x4 gf[2][MT][S];
                auto load_w = [&](int q, float (&w)[NTW][4]) {
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            if (WTG) {
                                if (m == 0) {
                                    const f32x4 wv = pinn_ld4(wtg + ((size_t)li * HP + (wave * NTW + j) * 16 + lr) * HP + 16 * q + 4 * lq);
                                    w[j][0] = wv[0]; w[j][1] = wv[1]; w[j][2] = wv[2]; w[j][3] = wv[3];
                                }
                            } else if (WTL) {
                                if (m == 0) {
                                    const f32x4 wv = pinn_ld4(WTs + (li * HP + (wave * NTW + j) * 16 + lr) * C::WT_LD + 16 * q + 4 * lq);
                                    w[j][0] = wv[0]; w[j][1] = wv[1]; w[j][2] = wv[2]; w[j][3] = wv[3];
                                }
                            } else {
                                w[j][m] = WPF ? wqall[WPF ? q : 0][j][m]
                                              : Wl[(16 * q + 4 * lq + m) * HP + (wave * NTW + j) * 16 + lr];
                            }
                        }
                };
                auto load_g = [&](int q, f32x4 (&gfr)[MT][S]) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int s = 0; s < S; ++s)
                            gfr[mt][s] = pinn_ld4(nxt + (s * T + mt * 16 + lr) * LDA + 16 * q + 4 * lq);
                };
#pragma unroll
                for (int i = 0; i < WAD; ++i)
                    if (i < NQD) load_w(i, wq[i]);
                load_g(0, gf[0]);
#pragma unroll
                for (int q = 0; q < NQD; ++q) {
                    PINN_SCHED_BARRIER();
                    if (q + WAD < NQD) load_w(q + WAD, wq[(q + WAD) % (WAD + 1)]);
                    if (q + 1 < NQD) load_g(q + 1, gf[(q + 1) & 1]);
                    if (!PINN_SCHED_IL) PINN_SCHED_BARRIER();
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int s = 0; s < S; ++s)
#pragma unroll
                                for (int j = 0; j < NTW; ++j)
                                    g[j][mt][s] = pinn_mfma16(wq[q % (WAD + 1)][j][m], gf[q & 1][mt][s][m], g[j][mt][s]);
                    if (q + 1 < NQD)
                        pinn_sched_interleave<4 * MT * S * NTW, MT * S + ((WTL || WTG) ? NTW : (WPF ? 0 : 4 * NTW))>();
                    PINN_SCHED_BARRIER();
                }
            }
            if (WGX && GZ_LATE) store_gz();
            PH(13)
            tsync();
            PH(14)
        };
        if constexpr (DWG) {
            for (int a = lh; a >= 1; --a) hidden_reverse(a, dW[0], accBr[0]);
        } else {
#pragma unroll
            for (int a = PINN_LHMAX; a >= 1; --a) {
                if (a <= lh) hidden_reverse(a, dW[a - 1], accBr[REGB && a < NBR ? a : 0]);
            }
        }
        {
            // first layer: db_0, dW1[n][c] += sum_pt gz0 x_c  (+ sum_pt gz_k when c == col_k)
            f32x4 gz[NTW][MT][S];
            act_reverse(0, gz, accBr[0]);
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                if (REGB) {
#pragma unroll
                    for (int c = 0; c < W1R; ++c) {
                        if (c < d) {
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) {
                                accW1r[REGB ? c : 0][j] += gz[j][mt][0] * xs_t[(mt * 16 + lr) * PINN_XS_LD + c];
#pragma unroll
                                for (int k = 0; k < ND; ++k)
                                    if (pinn_dir_has<pinn_dir_x(N2)>(SH::dir(A, k), c)) accW1r[REGB ? c : 0][j] += (SH::FIXED ? 1.0f : pinn_dir_coef<pinn_dir_x(N2)>(SH::dir(A, k), c)) * gz[j][mt][1 + k];
                            }
                        }
                    }
                }
                for (int c = REGB ? W1R : 0; c < d; ++c) {
                    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                    float old_e[4] = {0.f, 0.f, 0.f, 0.f};
                    if (PINN_BIAS_EARLY_LD) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) old_e[r] = accW1[(unit0(j) + r) * PINN_XS_LD + c];
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        v += gz[j][mt][0] * xs_t[(mt * 16 + lr) * PINN_XS_LD + c];
#pragma unroll
                        for (int k = 0; k < ND; ++k)
                            if (pinn_dir_has<pinn_dir_x(N2)>(SH::dir(A, k), c)) v += (SH::FIXED ? 1.0f : pinn_dir_coef<pinn_dir_x(N2)>(SH::dir(A, k), c)) * gz[j][mt][1 + k];
                    }
                    v = pinn_row_sum16_v4(v);
                    if (lr == 0) {
                        // (all four reads first: written element by element these are four LDS round trips in a row)
                        float old[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) old[r] = PINN_BIAS_EARLY_LD ? old_e[r] : accW1[(unit0(j) + r) * PINN_XS_LD + c];
#pragma unroll
                        for (int r = 0; r < 4; ++r) accW1[(unit0(j) + r) * PINN_XS_LD + c] = old[r] + v[r];
                    }
                }
            }
        }
        PH(15)
    }
    if (SKEW > 0 && team == 0)
        for (int i = 0; i < SKEW; ++i) tsync();
    FPB(3)
    if (TEAM_FLAGS) PINN_SYNC();       // (the arrival counter shares its LDS slot with `scal`, written below)
    PH_FLUSH

    if (!train) return;
    if (REGB) {
        // one row reduction for everything the lanes summed privately; each (wave, j, lq) owns its units. Plain LDS STORES: with
        // register accumulators nothing else writes these slots (they were zeroed in the prologue; the per-tile LDS adds belong to
        // the kernels without REGB and to the columns c >= W1R) -- as read-modify-writes these were ~20 dependent LDS round trips
        // at the end of every workgroup (phase clocks of the one-CU fit chunk: 6.1 K ticks of a 30 K tick pass over one tile)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
#pragma unroll
            for (int a = 0; a <= PINN_LHMAX; ++a) {
                if (a <= lh) {
                    const f32x4 t = pinn_row_sum16_v4(accBr[REGB && a < NBR ? a : 0][j]);
                    if (lr == 0) pinn_st4(accB + a * HP + unit0(j), t);
                }
            }
#pragma unroll
            for (int c = 0; c < W1R; ++c) {
                if (c < d) {
                    const f32x4 t = pinn_row_sum16_v4(accW1r[REGB ? c : 0][j]);
                    if (lr == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) accW1[(unit0(j) + r) * PINN_XS_LD + c] = t[r];
                    }
                }
            }
        }
    }
    FPB(4)
    // ---- write this workgroup's partial gradient ---------------------------------------------------------------
    // (two teams: ONE row per workgroup -- team 0 stores, team 1 adds on top behind a barrier)
    // the per-point sums of the point-stage lanes (tid < T: rows of wave 0) -> every lane of wave 0, by row and wave sums (as LDS slots
    // summed by thread 0 these were 48 serial LDS reads at the end of every workgroup: 5 K ticks)
    static_assert(T <= 64, "the point-stage lanes sit in wave 0");
    // (round 6: across the lanes in DOUBLE, one rounding to fp32 per row -- d loss / d b_L and d loss / d log_scale are sums of signed
    //  per-point terms that cancel, BASELINE config 4's to 1 / 850 of the sum of their magnitudes; with fp32 row sums the lane tree alone
    //  put 3 - 4e-6 relative onto that entry, tools/cfg4_bl_probe.py. Once per workgroup: the cost does not show)
    float tot_loss = 0.0f, tot_ls = 0.0f, tot_bl = 0.0f, tot_ic = 0.0f;
    if (wave == ptw) {
        double t64[4] = {pt_thread ? (double)sum_loss : 0.0, pt_thread ? (double)sum_ls : 0.0, pt_thread ? (double)sum_bl : 0.0,
                         pt_thread ? (double)sum_ic : 0.0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            t64[i] = pinn_row_sum16_f64(t64[i]);                 // (DPP on both halves; no LDS round trips: the one-launch fit chunk of the
            if (T > 16) t64[i] = pinn_rows_total_f64(t64[i]);    //  narrow nets passes here once per ITERATION)
        }
        tot_loss = (float)t64[0]; tot_ls = (float)t64[1]; tot_bl = (float)t64[2]; tot_ic = (float)t64[3];
    }
    (void)scal;
    PINN_SYNC();
    float* part = partials_ + (size_t)rowid * rstride_;
    for (int round = 0; round < (VWG ? 1 : TEAMS); ++round) {
        if (VWG || team == round) {
            const bool add = round > 0;
            auto put = [&](float* p, float v) { *p = add ? *p + v : v; };
#pragma unroll
            for (int l = 0; l < (DWG ? 0 : PINN_LHMAX); ++l) {
                if (l < lh) {
                    float* dst = part + A.off_wh + (size_t)l * A.hidden_stride;
#pragma unroll
                    for (int o = 0; o < NT; ++o)
#pragma unroll
                        for (int j = 0; j < NTW; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                put(dst + (o * 16 + lq * 4 + r) * HP + (wave * NTW + j) * 16 + lr, dW[l][o][j][r]);
                }
            }
            for (int i = tid; i < (lh + 1) * HP; i += NTHREADS) {
                const int a_ = i / HP, n = i % HP;
                const int dst = (a_ == 0) ? A.off_b1 + n : A.off_wh + (a_ - 1) * A.hidden_stride + HP * HP + n;
                put(part + dst, accB[i]);
            }
            for (int i = tid; i < HP * d; i += NTHREADS) put(part + i, accW1[(i / d) * PINN_XS_LD + (i % d)]);
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = pinn_row_sum16(accWL[j][r]);
                    if (lr == 0) put(part + A.off_wl + unit0(j) + r, v);
                }
            }
            if (tid == 64 * ptw) {                  // (lane 0 of the team's point-stage wave holds the totals)
                put(part + A.off_loss, tot_loss);
                put(part + A.off_ls, tot_ls);
                put(part + A.off_bl, tot_bl);
                if (!add)
                    for (int i = A.off_loss + 1; i < A.p_core; ++i) part[i] = 0.0f;
                if ((SPEC == 0 || PROG) && SH::mode(A) == PINN_MODE_STEP && SH::res_kind(A) == PINN_RES_PROGRAM) {
                    const int vbase = S + d + A.n_aux;
                    for (int k = 0; k < A.n_vars; ++k) {
                        float g = 0.0f;
                        for (int i = 0; i < T; ++i) g += padj[(vbase + k) * T + i];
                        put(part + A.off_extra + k, g);          // (two teams: the second one adds on top)
                    }
                }
                if (SPEC == 0 && A.ic_var1 > 0 && SH::mode(A) == PINN_MODE_STEP) {
                    part[A.off_extra + A.ic_var1 - 1] += tot_ic;
                }
            }
        }
        if (TEAMS2) { PINN_FENCE_BLOCK(); PINN_SYNC(); }
    }
    FPB(5)
}

template <int HP, int ND, int N2, int MT, int LHC, int ACTC, bool COMB = false, int VAR = 0>
PINN_GLOBAL void PINN_LAUNCH_BOUNDS2((PinnCfg<HP, ND, N2, MT>::NTHREADS * ((VAR & 256) ? 2 : 1)),
                                    (PinnCfg<HP, ND, N2, MT>::NW < 4 || (VAR & (2 | 256)) ? 2 : PINN_WAVES_PER_SIMD))
pinn_tile_kernel(const PinnKArgs A) {
    pinn_tile_body<HP, ND, N2, MT, LHC, ACTC, COMB, VAR>(A, A.params, A.partials, A.xs, A.aux, A.p_core);
}
