// pinn_duo_kernel.h -- the "two-team" form of the fused PINN tile kernel (training modes, widths <= 64).
//
// Same mathematics and lane maps as pinn_tile_kernel (pinn_kernel.h); different occupancy structure. One workgroup =
// TWO teams of NT waves (NT = HP/16, one wave per unit tile); each team streams its own 16-point tiles through the
// phase list below, and team 1 runs exactly ONE PHASE BEHIND team 0. Phases alternate between matrix work (MFMA) and
// vector work (activation jets, staging), so on every SIMD one wave is in an MFMA phase while its partner wave is in a
// VALU phase: the two pipes of the SIMD overlap without any intra-wave software pipelining, and every LDS / L2 latency
// of one team is covered by the other. The only synchronisation is one workgroup barrier per phase slot.
// Weight-gradient accumulators live in LDS shared by both teams (ds_add_f32), which also takes 48 VGPRs off every
// wave: 8 waves x <= 256 registers fit the CU.
//
//   phase list of one tile (LH hidden->hidden layers), NPH = 4*LH + 1:
//     0            V  layer-0 reverse of the PREVIOUS tile  +  first layer of this tile
//     2l-1, 2l     M  forward GEMM of hidden layer l  /  V  its activation jets          (l = 1..LH; last one + head dot)
//     2LH+1        V  point stage (ansatz, residual, adjoints) + reverse through the last layer + jets reverse LH
//     then for a = LH..1:   M  weight-grad + data-grad GEMMs of layer a   /   V  jets reverse of layer a-1  (a > 1)
#pragma once
#include "pinn_kernel.h"

template <int V> struct PinnIC { static constexpr int value = V; };

template <int HP_, int ND_, int N2_, int LH_>
struct PinnDuoCfg {
    static constexpr int HP = HP_, ND = ND_, N2 = N2_, LH = LH_;
    static constexpr int S = 1 + ND + N2;
    static constexpr int NT = HP / 16;                     // waves per team (one unit tile each)
    static constexpr int T = 16;
    static constexpr int LDA = HP + 8;
    static constexpr int NTHREADS = 2 * NT * 64;
    static constexpr int NPH = 4 * LH + 1;
    // shared LDS (floats)
    static constexpr int O_W1 = 0;
    static constexpr int O_B1 = O_W1 + HP * PINN_XS_LD;
    static constexpr int O_WL = O_B1 + HP;
    static constexpr int O_ACCB = O_WL + HP;
    static constexpr int O_ACCW1 = O_ACCB + (LH + 1) * HP;
    static constexpr int O_DW = O_ACCW1 + HP * PINN_XS_LD;
    static constexpr int O_SCAL = O_DW + LH * HP * HP;
    static constexpr int O_TEAM = O_SCAL + 2 * T * 4;
    // per-team LDS (floats)
    static constexpr int T_XS = 0;                          // [2][T][8]
    static constexpr int T_BUFA = T_XS + 2 * T * PINN_XS_LD;
    static constexpr int T_BUFB = T_BUFA + S * T * LDA;
    static constexpr int T_NET = T_BUFB + S * T * LDA;      // [NT][S][T]
    static constexpr int T_PREG = T_NET + NT * S * T;
    static constexpr int T_PADJ = T_PREG + PINN_MAX_REGS * T;
    static constexpr int TEAM_FLOATS = T_PADJ + PINN_MAX_REGS * T;
    static constexpr int SMEM_FLOATS = O_TEAM + 2 * TEAM_FLOATS;
    static constexpr bool FITS = SMEM_FLOATS * 4 <= 160 * 1024 && LH >= 1 && LH <= PINN_LHMAX && NT <= 4;
    // activation slab: lane-private float4s of the hidden activations 1..LH-1 (S each) and of layer 0 (value only)
    PINN_HOST_DEVICE static constexpr size_t slab_vec4_per_wg() { return (size_t)(LH + 1) * S * NTHREADS; }
};

template <int HP, int ND, int N2, int LH, int ACT>
PINN_GLOBAL void PINN_LAUNCH_BOUNDS2((PinnDuoCfg<HP, ND, N2, LH>::NTHREADS), (PinnDuoCfg<HP, ND, N2, LH>::NT == 4 ? 2 : 1))
pinn_duo_kernel(const PinnKArgs A) {
    using C = PinnDuoCfg<HP, ND, N2, LH>;
    constexpr int S = C::S, NT = C::NT, T = C::T, LDA = C::LDA, NTHREADS = C::NTHREADS, NPH = C::NPH, NQ = HP / 16;
    const int tid = PINN_TID, lane = tid & 63, wave_wg = tid >> 6;
    const int team = wave_wg / NT, wave = wave_wg % NT;          // wave = unit tile owned inside the team
    const int ttid = tid - team * NT * 64;                       // thread index inside the team
    const int lr = lane & 15, lq = lane >> 4;
    const int act = ACT, d = A.d;
    const int n0 = wave * 16 + 4 * lq;                           // first of this lane's 4 units

    PINN_SMEM(smem);
    float* W1s = smem + C::O_W1;
    float* b1s = smem + C::O_B1;
    float* WLs = smem + C::O_WL;
    float* accB = smem + C::O_ACCB;
    float* accW1 = smem + C::O_ACCW1;
    float* dWs = smem + C::O_DW;
    float* scal = smem + C::O_SCAL;
    float* tb = smem + C::O_TEAM + team * C::TEAM_FLOATS;
    float* xs_t = tb + C::T_XS;
    float* bufA = tb + C::T_BUFA;
    float* bufB = tb + C::T_BUFB;
    float* netp = tb + C::T_NET;
    float* pregs = tb + C::T_PREG;
    float* padj = tb + C::T_PADJ;

    for (int i = tid; i < HP * PINN_XS_LD; i += NTHREADS) {
        const int n = i / PINN_XS_LD, c = i % PINN_XS_LD;
        W1s[i] = (c < d) ? A.params[n * d + c] : 0.0f;
        accW1[i] = 0.0f;
    }
    for (int i = tid; i < HP; i += NTHREADS) { b1s[i] = A.params[A.off_b1 + i]; WLs[i] = A.params[A.off_wl + i]; }
    for (int i = tid; i < (LH + 1) * HP; i += NTHREADS) accB[i] = 0.0f;
    for (int i = tid; i < LH * HP * HP; i += NTHREADS) dWs[i] = 0.0f;
    for (int i = ttid; i < PINN_MAX_REGS * T; i += NT * 64) padj[i] = 0.0f;
    const float bL = A.params[A.off_bl];

    f32x4* slab = A.slab + (size_t)PINN_BID * C::slab_vec4_per_wg();
    auto slab_at = [&](int a, int s) PINN_INLINE_LAMBDA -> f32x4*  { return slab + ((size_t)a * S + s) * NTHREADS + tid; };
    auto wg_pt = [&](int m) PINN_INLINE_LAMBDA { return 2 * lq + (m & 1) + 8 * (m >> 1); };

    // tiles of this team: tile(k) = (k * nblk + bid) * 2 + team
    const long long ntiles = (A.n_points + T - 1) / T;
    auto tile_of = [&](long long k) PINN_INLINE_LAMBDA { return (k * PINN_NBLK + PINN_BID) * 2 + team; };
    long long my_tiles = 0;
    {
        const long long first = tile_of(0);
        if (first < ntiles) my_tiles = (ntiles - 1 - first) / (2LL * PINN_NBLK) + 1;
        if ((A.debug_flags & 1) && team == 1) my_tiles = 0;
    }
    // both teams run the same number of slots (one barrier each)
    const long long tiles_max = (ntiles - 1 - (long long)PINN_BID * 2) >= 0
                                    ? ((ntiles - 1 - (long long)PINN_BID * 2) / (2LL * PINN_NBLK) + 1) : 0;

    // ---- state carried from phase to phase (registers) -----------------------------------------------------------
    f32x4 acc[S];          // forward accumulators (M -> V) / data-gradient accumulators g (M -> V)
    f32x4 sv[S];           // saved jets of the activation being reversed
    // (a switch-in-loop phase machine keeps EVERY cross-phase variable live in every phase: the state is kept minimal.
    //  The top activation's saved jets go straight into `sv`; its values are recomputed from them when needed again.)
    f32x4 accWL = f32x4{0.f, 0.f, 0.f, 0.f};
    float sum_loss = 0.0f, sum_ls = 0.0f, sum_bl = 0.0f;
    float xpre[(T * PINN_XS_LD + NT * 64 - 1) / (NT * 64)];
    constexpr int NPRE = (T * PINN_XS_LD + NT * 64 - 1) / (NT * 64);
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = sv[s] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto fetch_points = [&](long long tile) PINN_INLINE_LAMBDA {
#pragma unroll
        for (int e = 0; e < NPRE; ++e) {
            const int i = ttid + e * NT * 64;
            const int pt = i / PINN_XS_LD, c = i % PINN_XS_LD;
            const long long g = tile * T + pt;
            xpre[e] = (i < T * PINN_XS_LD && c < d && tile < ntiles && g < A.n_points) ? A.xs[g * d + c] : 0.0f;
        }
    };
    auto store_points = [&](int buf) PINN_INLINE_LAMBDA {
#pragma unroll
        for (int e = 0; e < NPRE; ++e) {
            const int i = ttid + e * NT * 64;
            if (i < T * PINN_XS_LD) xs_t[buf * T * PINN_XS_LD + i] = xpre[e];
        }
    };
    if (my_tiles > 0) { fetch_points(tile_of(0)); store_points(0); fetch_points(tile_of(1)); }
    PINN_SYNC();

    // ---- helpers shared by the phases ---------------------------------------------------------------------------------
    // small LDS accumulators (biases, first layer): plain read-modify-write by one lane per row. Vector phases that
    // touch them (jets reverse) of the two teams never share a slot either (team 0 at 2LH+1+2i <-> team 1 in a GEMM phase;
    // phase 0 <-> the last reverse GEMM phase).
    auto lds_add4 = [&](float* dst, f32x4 v) PINN_INLINE_LAMBDA { pinn_st4(dst, pinn_ld4(dst) + v); };
    // jets reverse of activation a: g (in acc) -> gz; bias gradient (row sums over the 16 points) -> accB
    auto act_reverse = [&](int a, f32x4 (&gz)[S]) PINN_INLINE_LAMBDA {
        f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float gh1[S], sv1[S], gz1[S];
#pragma unroll
            for (int s = 0; s < S; ++s) { gh1[s] = acc[s][r]; sv1[s] = sv[s][r]; }
            pinn_jet_bwd<ND, N2>(gh1, sv1, act, gz1);
#pragma unroll
            for (int s = 0; s < S; ++s) gz[s][r] = gz1[s];
            bsum[r] = gz1[0];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bsum[r] = pinn_row_sum16(bsum[r]);
        if (lr == 0) lds_add4(accB + a * HP + n0, bsum);
    };
    // forward GEMM of one hidden layer: acc_s = W . h_s^T   (software-pipelined over the K quads)
    auto fwd_gemm = [&](const float* Wl, const float* hin) PINN_INLINE_LAMBDA {
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 wf[2], hf[2][S];
        auto load_q = [&](int q, f32x4& w, f32x4 (&h)[S]) PINN_INLINE_LAMBDA {
            w = pinn_ld4(Wl + (wave * 16 + lr) * HP + 16 * q + 4 * lq);
#pragma unroll
            for (int s = 0; s < S; ++s) h[s] = pinn_ld4(hin + (s * T + lr) * LDA + 16 * q + 4 * lq);
        };
        load_q(0, wf[0], hf[0]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + 1 < NQ) load_q(q + 1, wf[(q + 1) & 1], hf[(q + 1) & 1]);
            PINN_SCHED_BARRIER();
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int s = 0; s < S; ++s) acc[s] = pinn_mfma16(wf[q & 1][m], hf[q & 1][s][m], acc[s]);
            PINN_SCHED_BARRIER();
        }
    };
    // activation jets on the accumulators of hidden activation `a` (1..LH). Not the top one: h -> `hout`, saved jets -> slab.
    // Top one: saved jets stay in `sv` (first thing the reverse half needs), h feeds the last-layer dot right here.
    auto fwd_jets = [&](int a, const float* bl, float* hout, bool top) PINN_INLINE_LAMBDA {
        const f32x4 bias = pinn_ld4(bl + n0);
        f32x4 hv[S], svv[S];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float z[S], h[S];
#pragma unroll
            for (int s = 0; s < S; ++s) z[s] = acc[s][r];
            z[0] += bias[r];
            pinn_jet_fwd<ND, N2>(z, act, h);
#pragma unroll
            for (int s = 0; s < S; ++s) { hv[s][r] = h[s]; svv[s][r] = (s == 0) ? h[0] : z[s]; }
        }
        if (top) {
            const f32x4 wv = pinn_ld4(WLs + n0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                sv[s] = svv[s];
                float part = hv[s][0] * wv[0];
                part = fmaf(hv[s][1], wv[1], part); part = fmaf(hv[s][2], wv[2], part); part = fmaf(hv[s][3], wv[3], part);
                part = pinn_rows_sum(part);
                if (lq == 0) netp[(wave * S + s) * T + lr] = part;
            }
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) { pinn_st4(hout + (s * T + lr) * LDA + n0, hv[s]); *slab_at(a, s) = svv[s]; }
        }
    };
    // weight-grad + data-grad GEMMs of hidden linear layer li (0-based); h in `hb`, gz in `gb`; result g in acc
    auto bwd_gemms = [&](int li, const float* hb, const float* gb) PINN_INLINE_LAMBDA {
        const float* Wl = A.params + A.off_wh + (size_t)li * A.hidden_stride;
        {
            // The LDS accumulator is kept in fragment order [layer][wave][o][lane] (one float4 per lane and tile) and
            // updated with plain loads/stores: team 1 runs one phase behind team 0 and reverse-GEMM phases sit two
            // phases apart, so the two teams are never in this code in the same slot (slots end with a barrier).
            f32x4* dwf = reinterpret_cast<f32x4*>(dWs) + ((size_t)(li * NT + wave) * NT) * 64 + lane;
            f32x4 dw[NT];
#pragma unroll
            for (int o = 0; o < NT; ++o) dw[o] = dwf[o * 64];
            float bq[2][4], aq[2][NT][4];
            auto load_s = [&](int s, float (&b)[4], float (&a_)[NT][4]) PINN_INLINE_LAMBDA {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int row = (s * T + wg_pt(m)) * LDA;
                    b[m] = hb[row + wave * 16 + lr];
#pragma unroll
                    for (int o = 0; o < NT; ++o) a_[o][m] = gb[row + o * 16 + lr];
                }
            };
            load_s(0, bq[0], aq[0]);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (s + 1 < S) load_s(s + 1, bq[(s + 1) & 1], aq[(s + 1) & 1]);
                PINN_SCHED_BARRIER();
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int o = 0; o < NT; ++o) dw[o] = pinn_mfma16(aq[s & 1][o][m], bq[s & 1][m], dw[o]);
                PINN_SCHED_BARRIER();
            }
#pragma unroll
            for (int o = 0; o < NT; ++o) dwf[o * 64] = dw[o];
        }
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        float wq[2][4];
        f32x4 gf[2][S];
        auto load_q = [&](int q, float (&w)[4], f32x4 (&gfr)[S]) PINN_INLINE_LAMBDA {
#pragma unroll
            for (int m = 0; m < 4; ++m) w[m] = Wl[(16 * q + 4 * lq + m) * HP + wave * 16 + lr];
#pragma unroll
            for (int s = 0; s < S; ++s) gfr[s] = pinn_ld4(gb + (s * T + lr) * LDA + 16 * q + 4 * lq);
        };
        load_q(0, wq[0], gf[0]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + 1 < NQ) load_q(q + 1, wq[(q + 1) & 1], gf[(q + 1) & 1]);
            PINN_SCHED_BARRIER();
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int s = 0; s < S; ++s) acc[s] = pinn_mfma16(wq[q & 1][m], gf[q & 1][s][m], acc[s]);
            PINN_SCHED_BARRIER();
        }
    };
    // V phase in front of the GEMMs of layer a: gz_a -> `gb`; h_{a-1} recomputed from its saved jets -> `hb`
    // (not for the top layer: the forward copy of h_{LH-1} is still in LDS); sv <- saved jets of activation a-1
    auto stage_reverse = [&](int a, float* hb, float* gb, bool top) PINN_INLINE_LAMBDA {
        f32x4 gz[S];
        act_reverse(a, gz);
        if (a == 1) {
            sv[0] = *slab_at(0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int k = 0; k < ND; ++k) sv[1 + k][r] = pinn_dir_weight(W1s + (n0 + r) * PINN_XS_LD, A.dir_cols[k]);
#pragma unroll
                for (int k = 0; k < N2; ++k) sv[1 + ND + k][r] = 0.0f;
            }
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) sv[s] = *slab_at(a - 1, s);
        }
        f32x4 hv[S];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sv1[S], h[S];
#pragma unroll
            for (int s = 0; s < S; ++s) sv1[s] = sv[s][r];
            pinn_jet_recompute<ND, N2>(sv1, act, h);
#pragma unroll
            for (int s = 0; s < S; ++s) hv[s][r] = h[s];
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (!top) pinn_st4(hb + (s * T + lr) * LDA + n0, hv[s]);
            pinn_st4(gb + (s * T + lr) * LDA + n0, gz[s]);
        }
    };

    // ---- the phase machine ----------------------------------------------------------------------------------------
    // forward buffers alternate A,B,A,...: h_a lives in (a even ? bufA : bufB); the reverse GEMMs of layer a read
    // h_{a-1} from where the forward pass put it and gz_a from the other buffer
    auto hbuf = [&](int a) PINN_INLINE_LAMBDA -> float*  { return (a & 1) ? bufB : bufA; };
    auto run_phase = [&](auto PHC, long long k, bool has_prev, bool has_tile) PINN_INLINE_LAMBDA {
        constexpr int PH = decltype(PHC)::value;
        // a vector phase shares its SIMD with the other team's MFMA phase: give the vector wave issue priority (its VALU
        // instructions slot in between the partner's 32-cycle MFMAs; the MFMA wave only needs one issue slot per MFMA)
        constexpr bool IS_GEMM = (PH >= 1 && PH <= 2 * LH && (PH & 1) == 1) ||
                                 (PH >= 2 * LH + 2 && ((PH - (2 * LH + 2)) & 1) == 0);
        if (IS_GEMM) { PINN_SETPRIO(0); } else { PINN_SETPRIO(3); }
        const long long tile = tile_of(k);
        const long long base = tile * T;
        const int xb = (int)(k & 1);
        if constexpr (PH == 0) {
            if (has_prev) {
                // layer-0 reverse of the previous tile (gh_0 is in acc, its saved jets in sv, its points in the other
                // xs buffer): db_0, dW1
                f32x4 gz0[S];
                act_reverse(0, gz0);
                const float* xprev = xs_t + (xb ^ 1) * T * PINN_XS_LD;
                for (int c = 0; c < d; ++c) {
                    f32x4 v = gz0[0] * xprev[lr * PINN_XS_LD + c];
#pragma unroll
                    for (int kk = 0; kk < ND; ++kk)
                        if (pinn_dir_has(A.dir_cols[kk], c)) v += gz0[1 + kk];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = pinn_row_sum16(v[r]);
                        if (lr == 0) accW1[(n0 + r) * PINN_XS_LD + c] += t;
                    }
                }
            }
            if (has_tile) {
                const float* x = xs_t + xb * T * PINN_XS_LD + lr * PINN_XS_LD;
                f32x4 hv[S], svv[S];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + r;
                    float z[S], h[S];
                    float z0 = b1s[n];
                    for (int c = 0; c < d; ++c) z0 = fmaf(W1s[n * PINN_XS_LD + c], x[c], z0);
                    z[0] = z0;
#pragma unroll
                    for (int kk = 0; kk < ND; ++kk) z[1 + kk] = pinn_dir_weight(W1s + n * PINN_XS_LD, A.dir_cols[kk]);
#pragma unroll
                    for (int kk = 0; kk < N2; ++kk) z[1 + ND + kk] = 0.0f;
                    pinn_jet_fwd<ND, N2>(z, act, h);
#pragma unroll
                    for (int s = 0; s < S; ++s) { hv[s][r] = h[s]; svv[s][r] = (s == 0) ? h[0] : z[s]; }
                }
#pragma unroll
                for (int s = 0; s < S; ++s) pinn_st4(hbuf(0) + (s * T + lr) * LDA + n0, hv[s]);
                *slab_at(0, 0) = svv[0];
            }
        } else if constexpr (PH <= 2 * LH && (PH & 1) == 1) {
            constexpr int l = (PH + 1) / 2;                     // hidden layer 1..LH, input h_{l-1}
            fwd_gemm(A.params + A.off_wh + (size_t)(l - 1) * A.hidden_stride, hbuf(l - 1));
        } else if constexpr (PH <= 2 * LH) {
            constexpr int l = PH / 2;
            fwd_jets(l, A.params + A.off_wh + (size_t)(l - 1) * A.hidden_stride + HP * HP, hbuf(l), l == LH);
        } else if constexpr (PH == 2 * LH + 1) {
            // point stage, evaluated by every lane for ITS point (pt = lr; 4x redundant over lq, no LDS round trip)
            float net[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float v = (s == 0) ? bL : 0.0f;
#pragma unroll
                for (int w = 0; w < NT; ++w) v += netp[(w * S + s) * T + lr];
                net[s] = v;
            }
            PinnPointOut<ND, N2> po;
            PinnPointPre<ND, N2> ppre;
            pinn_point_prefetch<ND, N2>(A, A.params, base + lr, base + lr < A.n_points, pregs + lr, T, ppre, A.aux);
            const bool writer = (wave == 0 && lq == 0);
            // (residual PROGRAMS keep per-point registers in LDS and are run by the solo kernel; the host only sends
            //  affine residuals and external upstream gradients here)
            pinn_point_stage<ND, N2, false>(A, A.params, net, xs_t + xb * T * PINN_XS_LD + lr * PINN_XS_LD, base + lr,
                                     base + lr < A.n_points, pregs + lr, padj + lr, T, ppre, po);
            if (writer) { sum_loss += po.loss; sum_ls += po.g_ls; sum_bl += po.gnet[0]; }
            // stage the next tile's points (other xs buffer) and start fetching the one after
            store_points(xb ^ 1);
            fetch_points(tile_of(k + 2));
            // reverse through the last layer and the top activation
            const f32x4 wl = pinn_ld4(WLs + n0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sv1[S], h[S];
#pragma unroll
                for (int s = 0; s < S; ++s) sv1[s] = sv[s][r];
                pinn_jet_recompute<ND, N2>(sv1, act, h);           // top activations again (cheaper than carrying them)
#pragma unroll
                for (int s = 0; s < S; ++s) accWL[r] = fmaf(h[s], po.gnet[s], accWL[r]);
            }
#pragma unroll
            for (int s = 0; s < S; ++s) acc[s] = wl * po.gnet[s];
            stage_reverse(LH, hbuf(LH - 1), hbuf(LH), true);
        } else if constexpr (((PH - (2 * LH + 2)) & 1) == 0) {
            constexpr int a = LH - (PH - (2 * LH + 2)) / 2;     // GEMMs of hidden linear layer a (1..LH)
            bwd_gemms(a - 1, hbuf(a - 1), hbuf(a));
        } else {
            constexpr int a = LH - (PH - (2 * LH + 3)) / 2 - 1;  // V phase in front of the GEMMs of layer a (1..LH-1)
            stage_reverse(a, hbuf(a - 1), hbuf(a), false);
        }
    };

    // Straight-line phase sequence with a ONE-BARRIER SKEW between the teams: team 1 passes one extra barrier before its
    // first phase and team 0 one extra barrier after its last, so in every barrier interval team 0 runs phase p while
    // team 1 runs phase p-1 (the hardware barrier only counts arriving waves; which s_barrier instruction a wave sits at
    // does not matter). Every wave executes the same number of barriers: tiles_max * NPH + 2.
    PH_DECL
    const int NW = NT * 2;   // waves per workgroup (PH_FLUSH indexes by it)
    (void)NW;
    if (team == 1) PINN_SYNC();
    for (long long k = 0; k <= tiles_max; ++k) {
        const bool has_tile = k < my_tiles, has_prev = k > 0 && k <= my_tiles;
        run_phase(PinnIC<0>{}, k, has_prev, has_tile);
        PH(0)
        PINN_SYNC();
        PH(15)
        if (k == tiles_max) break;
#define PINN_DUO_PHASE(P)                                              \
        if constexpr (P < NPH) {                                       \
            if (has_tile) run_phase(PinnIC<P>{}, k, has_prev, has_tile); \
            PH(P < 14 ? P : 14)                                        \
            PINN_SYNC();                                               \
            PH(15)                                                     \
        }
        PINN_DUO_PHASE(1) PINN_DUO_PHASE(2) PINN_DUO_PHASE(3) PINN_DUO_PHASE(4) PINN_DUO_PHASE(5) PINN_DUO_PHASE(6)
        PINN_DUO_PHASE(7) PINN_DUO_PHASE(8) PINN_DUO_PHASE(9) PINN_DUO_PHASE(10) PINN_DUO_PHASE(11) PINN_DUO_PHASE(12)
        PINN_DUO_PHASE(13) PINN_DUO_PHASE(14) PINN_DUO_PHASE(15) PINN_DUO_PHASE(16)
#undef PINN_DUO_PHASE
    }
    if (team == 0) PINN_SYNC();
    {
        const int wave = wave_wg;      // PH_FLUSH writes one row per wave of the workgroup
        (void)wave;
        PH_FLUSH
    }

    // ---- write this workgroup's partial gradient ---------------------------------------------------------------
    float* part = A.partials + (size_t)PINN_BID * A.p_core;
    for (int i = tid; i < LH * HP * HP; i += NTHREADS) {
        // fragment order: i = (((l*NT + w)*NT + o)*64 + lane)*4 + r  ->  dW_l[o*16 + (lane>>4)*4 + r][w*16 + (lane&15)]
        const int r = i & 3, ln = (i >> 2) & 63, o = (i >> 8) % NT, w = ((i >> 8) / NT) % NT, l = (i >> 8) / (NT * NT);
        part[A.off_wh + (size_t)l * A.hidden_stride + (o * 16 + (ln >> 4) * 4 + r) * HP + w * 16 + (ln & 15)] = dWs[i];
    }
    for (int i = tid; i < (LH + 1) * HP; i += NTHREADS) {
        const int a = i / HP, n = i % HP;
        const int dst = (a == 0) ? A.off_b1 + n : A.off_wh + (a - 1) * A.hidden_stride + HP * HP + n;
        part[dst] = accB[i];
    }
    for (int i = tid; i < HP * d; i += NTHREADS) part[i] = accW1[(i / d) * PINN_XS_LD + (i % d)];
    // last-layer weights: both teams hold partial sums in registers
    float* wlsum = tb + C::T_BUFA;                                // free now
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float v = pinn_row_sum16(accWL[r]);
        if (lr == 0) wlsum[n0 + r] = v;
    }
    if (wave == 0 && lq == 0) { scal[(team * T + lr) * 4 + 0] = sum_loss; scal[(team * T + lr) * 4 + 1] = sum_ls; scal[(team * T + lr) * 4 + 2] = sum_bl; }
    PINN_SYNC();
    for (int i = tid; i < HP; i += NTHREADS)
        part[A.off_wl + i] = (smem + C::O_TEAM + C::T_BUFA)[i] + (smem + C::O_TEAM + C::TEAM_FLOATS + C::T_BUFA)[i];
    if (tid == 0) {
        float l0 = 0.0f, l1 = 0.0f, l2 = 0.0f;
        for (int i = 0; i < 2 * T; ++i) { l0 += scal[i * 4]; l1 += scal[i * 4 + 1]; l2 += scal[i * 4 + 2]; }
        part[A.off_loss] = l0;
        part[A.off_ls] = l1;
        part[A.off_bl] = l2;
        for (int i = A.off_loss + 1; i < A.p_core; ++i) part[i] = 0.0f;
    }
}
