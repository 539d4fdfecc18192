// pinn_chain_kernel.h -- the fused PINN step for 64-wide nets with WAVE-PRIVATE points ("chained" MFMA layers) and a
// weight-gradient wave beside every chain wave.
//
// Same arithmetic, arguments and outputs as pinn_tile_kernel (pinn_kernel.h: forward Taylor-mode jets -> ansatz -> residual ->
// reverse sweep; reference pydens/model_torch.py:437-460), different execution model. pinn_tile_kernel splits the UNITS of
// a layer over the waves of a workgroup, so every layer's activations cross the LDS and a workgroup barrier (two per layer and
// direction). Here a CHAIN wave owns 16 * MT POINTS and computes all 64 units of every layer for them:
//
//   * D = W . H^T puts units on the accumulator rows (row = 4 * lq + r) and points on the columns (lane & 15). The next layer's
//     B operand wants k on lq and the point on lane & 15 -- and the contraction index may be enumerated in any order as long
//     as both operands agree, so MFMA step (q, m) is given k = 16 q + 4 lq + m: the B operand of that step is component m of
//     accumulator tile q, IN PLACE. Activations never leave the registers between layers; the forward and data-gradient
//     GEMMs need no LDS traffic but the weight fragments (one ds_read_b128 per 4 * MT * S MFMAs) and no barrier at all.
//   * the weight-gradient GEMM contracts over POINTS, which sit on the wrong lane axis for both operands: gz_a and h_{a-1}
//     go through an LDS transpose ([point][unit], one stream at a time). A chain wave has no registers left for 3 x 64 x 64
//     accumulators and, alone on its SIMD, nothing to fill its vector-phase latencies with -- so each SIMD runs a PAIR: the
//     chain wave (waves 0-3) stages the transposes into a two-slot LDS ring and moves on; its WGRAD wave (waves 4-7, same
//     SIMD) picks them up, keeps the whole dW of every hidden layer in its accumulators and issues its MFMAs whenever the
//     chain wave waits for something. Flags in LDS, no s_barrier: LDS instructions of a CU execute in issue order, so
//     "data, then flag" by one wave and "flag, then data" by the other need no waiting, only program order.
//   * the four wgrad waves' sums meet once, in LDS, after the last tile.
//
// 64-wide, tanh, static depth, shape-specialised (PinnShape 1..3) training steps only -- the BASELINE config 2 / 4 kernels;
// everything else stays on pinn_tile_kernel. Lane map and partial-row layout are those of pinn_tile_kernel, so
// pinn_reduce_kernel and the host side do not know the difference (to them a workgroup = four "teams").
#pragma once
#include "pinn_kernel.h"

template <int ND_, int N2P_, int MT_, int LHC_>
struct PinnChainCfg {
    static constexpr int HP = 64, NT = 4, NP = 4, NW = 8, NTHREADS = 512, MT = MT_, TW = 16 * MT_, LHC = LHC_;
    static constexpr int S = pinn_ns(ND_, N2P_);
    static constexpr int LDW = HP + 4;              // LDS row stride of W_l[out][in]: b128 rows (forward) and b32 columns (data gradient) conflict-free
    static constexpr int LDT = HP + 4;              // ... of the [point][unit] transposes of the weight-gradient GEMM
    static constexpr int O_W = 0;                                   // W_l, l < LHC; after the last tile: dW sums
    static constexpr int O_W1 = O_W + LHC * HP * LDW;               // [PINN_XS_LD columns][unit], columns >= d zero
    static constexpr int O_B = O_W1 + HP * PINN_XS_LD;              // b1 | hidden biases
    static constexpr int O_WL = O_B + (LHC + 1) * HP;
    static constexpr int ACCB_W = (LHC + 1) * HP, ACCW1_W = HP * PINN_XS_LD;
    static constexpr int O_ACCB = O_WL + HP;                        // per pair: bias gradients [(LHC + 1)][HP]
    static constexpr int O_ACCW1 = O_ACCB + NP * ACCB_W;            // per pair: first-layer weight gradient [HP][PINN_XS_LD]
    static constexpr int O_ACCWL = O_ACCW1 + NP * ACCW1_W;          // per pair: last-layer weight gradient [HP]
    static constexpr int O_SCAL = O_ACCWL + NP * HP;                // per pair: loss, d/dlog_scale, d/dbL, -
    static constexpr int O_FLAG = O_SCAL + NP * 4;                  // per pair: stages published, stages consumed, -, - (ints)
    static constexpr int SLOT = 2 * 16 * LDT;                       // one stage: gz | h, [16 points][LDT] each
    static constexpr int SCR_W = 2 * SLOT;                          // per pair: a ring of two stages
    static constexpr int O_SCR = O_FLAG + NP * 4;
    static constexpr int SMEM_FLOATS = O_SCR + NP * SCR_W;
    static_assert(LHC * HP * HP <= LHC * HP * LDW && LHC * HP * HP <= NP * SCR_W, "dW sums reuse the weight block and the ring");
    // saved jets of one chain wave's tile (lane private f32x4): the value of activation 0, S jets of activations 1 .. LHC - 1
    // (the top activation never leaves the registers)
    static constexpr int SLAB_VEC4 = (1 + (LHC - 1) * S) * NT * MT * 64;
};

template <int ND, int N2, int MT, int LHC, bool COMB, int SPEC>
PINN_GLOBAL void PINN_LAUNCH_BOUNDS2(512, 2)
pinn_chain_kernel(const PinnKArgs A) {
    using C = PinnChainCfg<ND, N2, MT, LHC>;
    using SH = PinnShape<SPEC, ND>;
    using J = PinnJet<ND, N2, COMB>;
    constexpr int S = C::S, HP = C::HP, NT = C::NT, NP = C::NP, TW = C::TW, LDW = C::LDW, LDT = C::LDT, NTHREADS = C::NTHREADS;
    constexpr int NW = C::NW;
    constexpr int ACT = PINN_ACT_TANH;
    constexpr int DX = (SPEC == 1 || SPEC == 2) ? ND : PINN_MAX_INPUTS;      // input columns a lane keeps of its point
    static_assert(SPEC >= 1 && SPEC <= 3 && LHC >= 1 && pinn_n3(N2) == 0 && (MT == 1 || MT == 2) && ND >= 1 && ND <= 4,
                  "chain kernels: shape-specialised training steps, static depth, up to second order");
    const int tid = PINN_TID, lane = tid & 63, wave = pinn_wave_uniform(tid >> 6), lr = lane & 15, lq = lane >> 4;
    const int pair = wave & 3, role = wave >> 2;     // waves w and w + 4 land on the same SIMD: chain wave w, its wgrad wave w + 4
    const int vbid = PINN_BID * NP + pair, vnblk = PINN_NBLK * NP;
    const float* cw = A.comb_w;
    const int d = SH::d(A);

    PH_DECL
    PINN_SMEM(smem);
    float* Ws = smem + C::O_W;
    float* W1s = smem + C::O_W1;
    float* bs = smem + C::O_B;
    float* WLs = smem + C::O_WL;
    float* accB = smem + C::O_ACCB + pair * C::ACCB_W;
    float* accW1 = smem + C::O_ACCW1 + pair * C::ACCW1_W;
    float* accWL = smem + C::O_ACCWL + pair * HP;
    float* scal = smem + C::O_SCAL;
    int* flags = reinterpret_cast<int*>(smem + C::O_FLAG) + pair * 4;      // [0] stages published, [1] stages consumed
    float* scr = smem + C::O_SCR + pair * C::SCR_W;

    // ---- one-time staging: hidden weights (rows padded to LDW), the small layers, zeroed accumulators -------------------
    {
        // all loads of a thread first (the weights were last written by another launch's Adam: every batch is an L2 miss)
        constexpr int NV = LHC * HP * HP / 4 / NTHREADS;
        f32x4 wreg[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int i = tid + e * NTHREADS, l = i / (HP * HP / 4), n = (i / (HP / 4)) % HP, k4 = i % (HP / 4);
            wreg[e] = pinn_ld4(A.params + A.off_wh + (size_t)l * A.hidden_stride + n * HP + 4 * k4);
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int i = tid + e * NTHREADS, l = i / (HP * HP / 4), n = (i / (HP / 4)) % HP, k4 = i % (HP / 4);
            pinn_st4(Ws + (l * HP + n) * LDW + 4 * k4, wreg[e]);
        }
    }
    for (int i = tid; i < HP * PINN_XS_LD; i += NTHREADS) {
        const int c = i / HP, n = i % HP;                      // column-major: W1s[c][unit] (one b128 = four units of a column)
        W1s[i] = (c < d) ? A.params[n * d + c] : 0.0f;
    }
    for (int i = tid; i < (LHC + 1) * HP; i += NTHREADS) {
        const int a = i / HP, n = i % HP;
        bs[i] = (a == 0) ? A.params[A.off_b1 + n] : A.params[A.off_wh + (size_t)(a - 1) * A.hidden_stride + HP * HP + n];
    }
    for (int i = tid; i < HP; i += NTHREADS) WLs[i] = A.params[A.off_wl + i];
    for (int i = tid; i < C::O_SCR - C::O_ACCB; i += NTHREADS) smem[C::O_ACCB + i] = 0.0f;      // accumulators, scalars, flags
    const float bL = A.params[A.off_bl];
    const long long ntiles = A.tile_end;              // tiles (of TW points) [A.tile_begin, A.tile_end) belong to this launch

    if (A.pre.n_ops > 0) {
        // x-only pre-pass (source term of the residual) for the points of this pair's own tiles, both waves, 128 / TW tiles per
        // sweep; its registers live in the ring (unused before the first tile) whenever they fit
        // (round 6: the pre-pass is fp64 and the host only hands it to kernels that report LDS room for its double registers -- this
        //  experiment kernel reports none, so the separate launch runs and this block stays a fallback with private registers)
        const int ptid = role * 64 + lane;
        for (long long tile = A.tile_begin + vbid + (long long)(ptid / TW) * vnblk; tile < ntiles; tile += (long long)(128 / TW) * vnblk) {
            const long long gi = tile * TW + ptid % TW;
            if (gi < A.n_points) pinn_prepass_point_private(A.pre, A.pre_consts64, A.xs + gi * d, d, A.aux, A.n_points, gi);
        }
        PINN_FENCE_BLOCK();
    }
    PINN_SYNC();

    if (role == 1) {
        // ================================ the weight-gradient wave ========================================================
    // the wgrad wave's persistent accumulators: the whole dW of every hidden layer; element r of dW[l][o][j] is
    // d loss / d W_l[16 o + 4 lq + r][16 j + lr], summed over the pair's points
    f32x4 dW[LHC][NT][NT];
#pragma unroll
    for (int l = 0; l < LHC; ++l)
#pragma unroll
        for (int o = 0; o < NT; ++o)
#pragma unroll
            for (int j = 0; j < NT; ++j) dW[l][o][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // dW[out][in] += sum_{s,pt} gz_s[out][pt] h_s[in][pt] -- A = gz^T, B = h, column reads of the stage's transposes; MFMA
        // k-slot (lq, m) <-> point 4 lq + m (conflict-free). Stage k of the pair sits in ring slot k & 1.
        int kst = 0;
        for (long long tile = A.tile_begin + vbid; tile < ntiles; tile += vnblk) {
#pragma unroll
            for (int a = LHC; a >= 1; --a) {
#pragma unroll
                for (int ms = 0; ms < MT * S; ++ms) {
                    if (A.debug_flags & 32) continue;                      // (timing experiments: no weight gradient at all)
                    while (pinn_flag_load(flags + 0) < kst + 1) PINN_SPIN_PAUSE();
                    PINN_WAVE_SYNC();
                    const float* gb = scr + (kst & 1) * C::SLOT;
                    const float* hb = gb + 16 * LDT;
                    float aq[NT][4], bq[NT][4];
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int o = 0; o < NT; ++o) {
                            aq[o][m] = gb[(4 * lq + m) * LDT + 16 * o + lr];
                            bq[o][m] = hb[(4 * lq + m) * LDT + 16 * o + lr];
                        }
                    ++kst;
                    pinn_flag_publish(flags + 1, kst, lane == 0);          // the slot is free again (reads execute before this write)
                    if (A.debug_flags & 8) continue;                       // (timing experiments: stages consumed, not multiplied)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int o = 0; o < NT; ++o)
#pragma unroll
                            for (int j = 0; j < NT; ++j)
                                dW[a - 1][o][j] = pinn_mfma16(aq[o][m], bq[j][m], dW[a - 1][o][j]);
                }
            }
        }
        // the four wgrad waves' sums meet in LDS: every wave is done with the weights and the ring (first barrier), waves 0 / 1
        // store their dW into the two blocks, waves 2 / 3 add on top
        PINN_SYNC();
        for (int round = 0; round < 2; ++round) {
            if ((pair >> 1) == round) {
                float* dst = smem + ((pair & 1) ? C::O_SCR : C::O_W);
#pragma unroll
                for (int l = 0; l < LHC; ++l)
#pragma unroll
                    for (int o = 0; o < NT; ++o) {
                        // second round: read-add-write, the 16 reads of a tile row in ONE batch (written element by element the
                        // compiler serialises 192 LDS round trips per lane; ds_add_f32 instead measured 5x slower still:
                        // LDS float atomics retire a few lanes per clock)
                        float old[NT][4];
#pragma unroll
                        for (int j = 0; j < NT; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                old[j][r] = (round > 0) ? dst[l * HP * HP + (16 * o + 4 * lq + r) * HP + 16 * j + lr] : 0.0f;
#pragma unroll
                        for (int j = 0; j < NT; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                dst[l * HP * HP + (16 * o + 4 * lq + r) * HP + 16 * j + lr] = old[j][r] + dW[l][o][j][r];
                    }
            }
            PINN_SYNC();
        }
    } else {
    // ==================================== the chain wave =================================================================
    PINN_SETPRIO(2);                                   // the critical path: its partner fills the gaps
    float sum_loss = 0.0f, sum_ls = 0.0f, sum_bl = 0.0f;
    int kst = 0;                                       // stages published so far

    const PinnRows slab = pinn_rows(A.slab + (size_t)vbid * C::SLAB_VEC4, C::SLAB_VEC4 * 16);
    auto slab_row = [&](int a, int s, int t, int mt) -> int {          // byte offset of a 64-lane row of f32x4
        const int slot = (a == 0) ? 0 : 1 + (a - 1) * S + s;
        return ((slot * NT + t) * MT + mt) * 64 * 16;
    };
    // v[t][r] belongs to unit 16 t + 4 lq + r; dst[unit * stride] += its row sum (all 16 sums of a lane in one batch)
    auto rowsum_add16 = [&](float* dst, int stride, const f32x4 (&v)[NT]) {
        float w[4 * NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) w[4 * t + r] = v[t][r];
        pinn_row_sum16_n<4 * NT>(w);
        if (lr == 0) {
            // (plain read-add-write, all reads first: these accumulators belong to this wave alone)
            float old[4 * NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) old[4 * t + r] = dst[(16 * t + 4 * lq + r) * stride];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(16 * t + 4 * lq + r) * stride] = old[4 * t + r] + w[4 * t + r];
        }
    };

    // the points of a tile are fetched one tile ahead (every lane keeps the columns of ITS point: lanes lq = 0..3 of a column
    // hold four copies, which is what the first layer and the point stage want)
    float xn[MT][DX];
    auto fetch_points = [&](long long tile) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const long long g = tile * TW + mt * 16 + lr;
            const bool ok = tile < ntiles && g < A.n_points;
#pragma unroll
            for (int c = 0; c < DX; ++c) xn[mt][c] = (ok && c < d) ? A.xs[g * d + c] : 0.0f;
        }
    };
    fetch_points(A.tile_begin + vbid);
    PH(11)

    for (long long tile = A.tile_begin + vbid; tile < ntiles; tile += vnblk) {
        const long long base = tile * TW;
        float x[MT][DX];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int c = 0; c < DX; ++c) x[mt][c] = xn[mt][c];
        fetch_points(tile + vnblk);
        PinnPointPre<ND, N2> ppre[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const long long g = base + mt * 16 + lr;
            pinn_point_prefetch<ND, N2, SPEC>(A, A.params, g, g < A.n_points, nullptr, 0, ppre[mt], A.aux);
        }

        PH(0)
        // ---- (1) first layer on the VALU: z0 = W1 x + b1, z_k = W1[:, k], z_kk = 0 ----------------------------------------
        f32x4 h[NT][MT][S];            // jets of the current activation = B operand of the next GEMM, in place
        f32x4 sv[NT][MT][S];           // saved form of the TOP activation (value, z_k, z_kk), filled by the last hidden layer
        {
            // all LDS reads first (one latency), then the arithmetic
            f32x4 b1v[NT], wc[NT][DX < 4 ? DX : 4];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                b1v[t] = pinn_ld4(bs + 16 * t + 4 * lq);
#pragma unroll
                for (int c = 0; c < (DX < 4 ? DX : 4); ++c) wc[t][c] = pinn_ld4(W1s + c * HP + 16 * t + 4 * lq);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 z0 = b1v[t];
#pragma unroll
                    for (int c = 0; c < (DX < 4 ? DX : 4); ++c) z0 += wc[t][c] * x[mt][c];
                    if (DX > 4 && d > 4) {
#pragma unroll
                        for (int c = 4; c < DX; ++c) z0 += pinn_ld4(W1s + (c < DX ? c : 0) * HP + 16 * t + 4 * lq) * x[mt][c < DX ? c : 0];
                    }
                    f32x4 v0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float z[S], hh[S];
                        z[0] = z0[r];
#pragma unroll
                        for (int k = 0; k < ND; ++k) z[1 + k] = wc[t][k < (DX < 4 ? DX : 4) ? k : 0][r];
#pragma unroll
                        for (int s = 1 + ND; s < S; ++s) z[s] = 0.0f;
                        pinn_jet_fwd<ND, N2, COMB>(z, ACT, hh, cw);
#pragma unroll
                        for (int s = 0; s < S; ++s) h[t][mt][s][r] = hh[s];
                        v0[r] = hh[0];
                    }
                    if (!(A.debug_flags & 128)) pinn_rows_st4(slab, lane * 16, slab_row(0, 0, t, mt), v0);          // z_k = W1[:, k] and z_kk = 0 are rebuilt in the reverse half
                }
            }
        }
        PH(1)
        // ---- (2) hidden layers: Z^T = W H^T, B operand = the previous layer's accumulators in place ------------------------
#pragma unroll
        for (int li = 0; li < LHC; ++li) {
            const float* Wl = Ws + li * HP * LDW;
            f32x4 acc[NT][MT][S];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int s = 0; s < S; ++s) acc[t][mt][s] = f32x4{0.f, 0.f, 0.f, 0.f};
            {
                f32x4 wf[2][NT];
                auto load_w = [&](int q, f32x4 (&w)[NT]) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) w[t] = pinn_ld4(Wl + (16 * t + lr) * LDW + 16 * q + 4 * lq);
                };
                load_w(0, wf[0]);
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    PINN_SCHED_BARRIER();
                    if (q + 1 < NT) load_w(q + 1, wf[(q + 1) & 1]);
                    if (!PINN_SCHED_IL) PINN_SCHED_BARRIER();
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int s = 0; s < S; ++s)
#pragma unroll
                                for (int t = 0; t < NT; ++t)
                                    acc[t][mt][s] = pinn_mfma16(wf[q & 1][t][m], h[q][mt][s][m], acc[t][mt][s]);
                    if (q + 1 < NT) pinn_sched_interleave<4 * MT * S * NT, NT>();
                    PINN_SCHED_BARRIER();
                }
            }
            PH(2)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 bias = pinn_ld4(bs + (li + 1) * HP + 16 * t + 4 * lq);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 svv[S];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float z[S], hh[S];
#pragma unroll
                        for (int s = 0; s < S; ++s) z[s] = acc[t][mt][s][r];
                        z[0] += bias[r];
                        pinn_jet_fwd<ND, N2, COMB>(z, ACT, hh, cw);
#pragma unroll
                        for (int s = 0; s < S; ++s) { h[t][mt][s][r] = hh[s]; svv[s][r] = (s == 0) ? hh[0] : z[s]; }
                    }
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        if (li + 1 == LHC) sv[t][mt][s] = svv[s];
                        else if (!(A.debug_flags & 128)) pinn_rows_st4(slab, lane * 16, slab_row(li + 1, s, t, mt), svv[s]);
                    }
                }
            }
            PH(3)
        }

        // ---- (3) last layer (out = 1): net_s[pt] = WL . h_s[pt] ; every lane of a column gets its point's sums ---------------
        f32x4 wlv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) wlv[t] = pinn_ld4(WLs + 16 * t + 4 * lq);
        float net[MT][S];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float acc1 = 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 hv = h[t][mt][s];
                    acc1 = fmaf(hv[0], wlv[t][0], acc1); acc1 = fmaf(hv[1], wlv[t][1], acc1);
                    acc1 = fmaf(hv[2], wlv[t][2], acc1); acc1 = fmaf(hv[3], wlv[t][3], acc1);
                }
                net[mt][s] = acc1;
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < S; ++s) net[mt][s] = pinn_rows_sum(net[mt][s]) + (s == 0 ? bL : 0.0f);

        PH(4)
        // ---- (4) ansatz + residual + their reverse: the four lanes of a column all do their point (no exchange needed
        //          afterwards); the sums count it once ---------------------------------------------------------------------
        float gnet[MT][S];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const long long g = base + mt * 16 + lr;
            PinnPointOut<ND, N2> po;
            pinn_point_stage<ND, N2, false, COMB, SPEC>(A, A.params, net[mt], x[mt], g, g < A.n_points, nullptr, nullptr, 0, ppre[mt], po);
#pragma unroll
            for (int s = 0; s < S; ++s) gnet[mt][s] = po.gnet[s];
            if (lq == 0) { sum_loss += po.loss; sum_ls += po.g_ls; sum_bl += po.gnet[0]; }
        }

        PH(5)
        // ---- (5) reverse through the last layer: gh_s = gnet_s * WL ; dWL += sum gnet_s h_s ---------------------------------
        f32x4 g[NT][MT][S];
        {
            f32x4 awl[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                awl[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        g[t][mt][s] = wlv[t] * gnet[mt][s];
                        awl[t] += h[t][mt][s] * gnet[mt][s];
                    }
            }
            rowsum_add16(accWL, 1, awl);
        }

        // saved jets of unit tile t of activation a (< LHC) from the slab; activation 0 keeps its value only
        auto load_saved = [&](int a, int t, f32x4 (&dst)[MT][S]) {
            if (A.debug_flags & 64) {                       // (timing experiments: no slab reads)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int s = 0; s < S; ++s) dst[mt][s] = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
                return;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (a == 0) {
                    dst[mt][0] = pinn_rows_ld4(slab, lane * 16, slab_row(0, 0, t, mt));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int k = 0; k < ND; ++k) dst[mt][1 + k][r] = W1s[k * HP + 16 * t + 4 * lq + r];
#pragma unroll
                        for (int s = 1 + ND; s < S; ++s) dst[mt][s][r] = 0.0f;
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < S; ++s) dst[mt][s] = pinn_rows_ld4(slab, lane * 16, slab_row(a, s, t, mt));
                }
            }
        };
        // gz = jet-reverse(gh, saved) of one unit tile, in place; returns sum_pt-partial of gz_0 (the bias gradient's lane share)
        auto tile_reverse = [&](f32x4 (&gg)[MT][S], const f32x4 (&svv)[MT][S]) -> f32x4 {
            f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float gh1[S], sv1[S], gz1[S];
#pragma unroll
                    for (int s = 0; s < S; ++s) { gh1[s] = gg[mt][s][r]; sv1[s] = svv[mt][s][r]; }
                    pinn_jet_bwd<ND, N2, COMB>(gh1, sv1, ACT, gz1, cw);
#pragma unroll
                    for (int s = 0; s < S; ++s) gg[mt][s][r] = gz1[s];
                    bsum[r] += gz1[0];
                }
            return bsum;
        };
        // activation a < LHC: its saved jets stream through a two-tile window (the window of tile 0 / 1 is already in flight)
        f32x4 win[2][MT][S];
        auto act_reverse_streamed = [&](int a) {
            f32x4 bsums[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                bsums[t] = tile_reverse(g[t], win[t & 1]);
                if (t + 2 < NT) load_saved(a, t + 2, win[t & 1]);
            }
            rowsum_add16(accB + a * HP, 1, bsums);
        };
        // one stage of the weight-gradient GEMM for the partner: gz_s and stream s of h_{a-1} (recomputed from its saved form,
        // as pinn_jet_recompute, one stream) of row tile mt, transposed into ring slot kst & 1
        auto publish_stage = [&](int mt, int s) {
            if (A.debug_flags & 32) return;
            const int seen = (kst >= 2) ? pinn_flag_load(flags + 1) : 0;          // asked for early, needed after the arithmetic
            f32x4 hp[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = sv[t][mt][0][r];
                    float d1, d2;
                    pinn_act_d12(v, ACT, d1, d2);
                    float o;
                    if (s == 0) o = v;
                    else if (s <= ND) o = d1 * sv[t][mt][s][r];
                    else {
                        o = d1 * sv[t][mt][s][r];
#pragma unroll
                        for (int k = 0; k < ND; ++k)
                            if (J::has2(k) && J::idx2(k) == s) o += d2 * J::w(k, cw) * sv[t][mt][1 + k][r] * sv[t][mt][1 + k][r];
                    }
                    hp[t][r] = o;
                }
            }
            if (kst >= 2 && seen < kst - 1 && !(A.debug_flags & 16)) {      // (16, timing experiments: never wait for the ring)
                while (pinn_flag_load(flags + 1) < kst - 1) PINN_SPIN_PAUSE();        // the slot's previous stage not yet read
            }
            PINN_WAVE_SYNC();
            float* gb = scr + (kst & 1) * C::SLOT;
            float* hb = gb + 16 * LDT;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                pinn_st4(gb + lr * LDT + 16 * t + 4 * lq, g[t][mt][s]);
                pinn_st4(hb + lr * LDT + 16 * t + 4 * lq, hp[t]);
            }
            ++kst;
            pinn_flag_publish(flags + 0, kst, lane == 0);
        };

        PH(6)
        // ---- (6) reverse through the hidden layers ---------------------------------------------------------------------------
#pragma unroll
        for (int a = LHC; a >= 1; --a) {
            if (a == LHC) {
                // top: the saved jets are in registers; those of the activation below replace them tile by tile
                f32x4 bsums[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    bsums[t] = tile_reverse(g[t], sv[t]);
                    load_saved(a - 1, t, sv[t]);
                }
                rowsum_add16(accB + a * HP, 1, bsums);
            } else {
                // (sv: all saved jets of activation a - 1 for the stages below; the window streams those of activation a)
#pragma unroll
                for (int t = 0; t < NT; ++t) load_saved(a - 1, t, sv[t]);
                act_reverse_streamed(a);
            }
            PH(7)
            // weight gradient of layer a - 1: stages for the partner (it multiplies while this wave goes on). Before the data
            // gradient: the saved jets the stages read are dead afterwards, and the GEMM below runs on g, gn and the window only
#pragma unroll
            for (int ms = 0; ms < MT * S; ++ms) publish_stage(ms / S, ms % S);
            // the window of the next activation reverse: its first two unit tiles
            if (a - 1 >= 0) {
                load_saved(a - 1, 0, win[0]);
                load_saved(a - 1, 1, win[1]);
            }
            PH(8)
            // data gradient: GH^T[in][pt] = sum_out W[out][in] gz[out][pt] -- A = column reads of W_l (b32, conflict-free at
            // LDW = 68), B = gz in place
            const float* Wl = Ws + (a - 1) * HP * LDW;
            f32x4 gn[NT][MT][S];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int s = 0; s < S; ++s) gn[t][mt][s] = f32x4{0.f, 0.f, 0.f, 0.f};
            {
                float wq[2][NT][4];
                auto load_q = [&](int q, float (&w)[NT][4]) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int m = 0; m < 4; ++m) w[t][m] = Wl[(16 * q + 4 * lq + m) * LDW + 16 * t + lr];
                };
                load_q(0, wq[0]);
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    PINN_SCHED_BARRIER();
                    if (q + 1 < NT) load_q(q + 1, wq[(q + 1) & 1]);
                    if (!PINN_SCHED_IL) PINN_SCHED_BARRIER();
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int s = 0; s < S; ++s)
#pragma unroll
                                for (int t = 0; t < NT; ++t)
                                    gn[t][mt][s] = pinn_mfma16(wq[q & 1][t][m], g[q][mt][s][m], gn[t][mt][s]);
                    if (q + 1 < NT) pinn_sched_interleave<4 * MT * S * NT, 2 * NT>();
                    PINN_SCHED_BARRIER();
                }
            }
            PH(9)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int s = 0; s < S; ++s) g[t][mt][s] = gn[t][mt][s];
        }
        // ---- (7) first layer: db_0, dW1[n][c] += sum_pt gz0 x_c  (+ sum_pt gz_k when c == k) --------------------------------
        act_reverse_streamed(0);
#pragma unroll
        for (int c = 0; c < DX; ++c) {
            if (c < d) {
                f32x4 v[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    v[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        v[t] += g[t][mt][0] * x[mt][c];
                        if (c < ND) v[t] += g[t][mt][1 + (c < ND ? c : 0)];
                    }
                }
                rowsum_add16(accW1 + c, PINN_XS_LD, v);
            }
        }
        PH(10)
    }
    {
        const float l0 = pinn_row_sum16(sum_loss), l1 = pinn_row_sum16(sum_ls), l2 = pinn_row_sum16(sum_bl);
        if (lane == 0) { scal[pair * 4 + 0] = l0; scal[pair * 4 + 1] = l1; scal[pair * 4 + 2] = l2; }
    }
    PINN_SETPRIO(0);
    PINN_SYNC();            // (the three barriers of the wgrad waves' epilogue)
    PINN_SYNC();
    PINN_SYNC();
    }   // chain wave

    // ---- the workgroup's partial gradient row ---------------------------------------------------------------------------------
    const float* sumW0 = smem + C::O_W;
    const float* sumW1 = smem + C::O_SCR;
    float* part = A.partials + (size_t)PINN_BID * A.p_core;
    for (int i = tid; i < LHC * HP * HP; i += NTHREADS)
        part[A.off_wh + (size_t)(i / (HP * HP)) * A.hidden_stride + i % (HP * HP)] = sumW0[i] + sumW1[i];
    auto over_pairs = [&](int off, int stride, int i) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < NP; ++w) v += smem[off + w * stride + i];
        return v;
    };
    for (int i = tid; i < (LHC + 1) * HP; i += NTHREADS) {
        const int a_ = i / HP, n = i % HP;
        const int dst = (a_ == 0) ? A.off_b1 + n : A.off_wh + (a_ - 1) * A.hidden_stride + HP * HP + n;
        part[dst] = over_pairs(C::O_ACCB, C::ACCB_W, i);
    }
    for (int i = tid; i < HP * d; i += NTHREADS) part[i] = over_pairs(C::O_ACCW1, C::ACCW1_W, (i / d) * PINN_XS_LD + (i % d));
    for (int i = tid; i < HP; i += NTHREADS) part[A.off_wl + i] = over_pairs(C::O_ACCWL, HP, i);
    if (tid == 0) {
        part[A.off_loss] = over_pairs(C::O_SCAL, 4, 0);
        part[A.off_ls] = over_pairs(C::O_SCAL, 4, 1);
        part[A.off_bl] = over_pairs(C::O_SCAL, 4, 2);
        for (int i = A.off_loss + 1; i < A.p_core; ++i) part[i] = 0.0f;
    }
    PH(12)
    PH_FLUSH
}
