// pinn_wgrad_kernel.h -- streamed weight-gradient GEMMs of the hidden->hidden layers for widths >= 128.
//
// Replaces, for those widths, the dW part of `loss.backward()` (pydens/model_torch.py:460):
//     dW_a[out][in] = sum over points and streams of  gz_a,s[pt][out] * h_{a-1},s[pt][in],     a = 1 .. LH.
// Why its own launch: a workgroup's dW (LH x HP x HP floats: 1.3 MB at 5 x 256 x 256) fits neither the register file
// (512 KB per CU) nor the LDS, so the fused tile kernel used to read-modify-write one 16 x HP tile row of it per 16-point
// tile in its partial-gradient buffer -- 21 GB of traffic per 131 072-point launch of BASELINE config 5 from a 335 MB working
// set, 44 % of the kernel. Here ONE layer's whole HP x HP accumulator lives in registers
// (8 waves x 128 x 64 outputs = 128 registers per lane at HP = 256) for ALL the tiles of the workgroup and is written once
// per layer; the operands stream in from HBM exactly once: the tile kernel (VAR 128, "WGX") leaves gz_a and the saved jets
// of every tile there, lane-private and coalesced (16 B per lane, 1 KB per wave instruction), 44 B per point, layer and
// unit instead of the 512 KB-per-tile-row read-modify-write.  h_{a-1} is rebuilt from the saved jets on the fly
// (a handful of VALU instructions per element against 2 x HP MFMA flops).
//
// Work split: workgroup b takes tiles b, b + grid, ... of the launch's tile range for every layer in turn.  Per
// (tile, row tile mt, stream s) -- one "stage", K = 16 points -- the 512 threads drop gz and h as unit-major rows
// [HP][16 + 4] into LDS (ds_write_b32, conflict-free), double-buffered, one barrier per stage; every wave then reads its
// A / B fragments as ds_read_b128 along K (the MFMA k-slot -> point map is a free bijection, shared by both operands) and
// issues AM x BN x 4 v_mfma_f32_16x16x4_f32 (exact fp32).  Wave grid 2 x 4 over the HP x HP output.
#pragma once
#include "pinn_kernel.h"

template <int HP, int ND, int N2, int MT = 1, bool SPLIT_ = false>
struct PinnWgCfg {
    using C = PinnCfg<HP, ND, N2, MT>;
    static constexpr bool SPLIT = SPLIT_;
    static constexpr int S = C::S, NW = C::NW, NTW = C::NTW, NTHREADS = C::NTHREADS;
    static_assert(NW == 8, "the streamed weight-gradient kernel is built for the 8-wave widths (HP >= 128)");
    static constexpr int KC = 16;                    // K rows per stage: the 16 points of one (tile, mt, stream)
    static constexpr int LDK = KC + 4;               // row stride of the unit-major operand buffers: rows 4 units apart land
                                                     // 16 banks apart (ds_write_b32 of lanes lq, lq + 1), b128 rows stay aligned
    static constexpr int WM = 2, WN = 4;             // wave grid over the output
    // width 512 (round 6): the HP x HP accumulator is 2 048 registers per lane -- the output goes in NBLK x NBLK blocks of HB x HB = 256 x 256,
    // one after the other; a block pass stages the HB units of gz_a of its row block and the HB units of h_{a-1} of its column block (every
    // operand is read from HBM NBLK times: twice)
    static constexpr int HB = HP > 256 ? 256 : HP, NBLK = HP / HB;
    static constexpr int AM = HB / 16 / WM, BN = HB / 16 / WN;    // 16 x 16 output tiles per wave (8 x 4 at 256 / 512, 4 x 2 at 128)
    static constexpr int OPER = HB * LDK;            // one operand buffer (floats)
    // split-bf16 form (round 3): TWO stages (K = 32 (stream, point) slots) per MFMA step; an operand buffer = three bf16 planes
    // of [HP units][32 k] = rows of 64 bytes, k slot 2 * point + stage: the two stages' values of a thread share one dword,
    // written by one ds_write_b32 per plane; 16-byte chunk index XORed with 3 * ((unit >> 2) & 1): ds_read_b128 of the fragments
    // conflict-free, the writes two-way (free for ds_write_b32). Two such buffer pairs (double buffering) fit at width 128
    // (96 KB), one at width 256 (96 KB: the staging of the next pair then waits for the MFMAs of this one).
    static_assert(!SPLIT || NBLK == 1, "split-bf16 weight gradients: widths up to 256");
    static constexpr int SP_PLANE_F = HP * 16;       // floats of one plane
    static constexpr int SP_OPER_F = 3 * SP_PLANE_F; // one operand (hi, mid, lo)
    static constexpr int SP_NBUF = (2 * 2 * SP_OPER_F * 4 + HP * PINN_XS_LD * 4 <= 160 * 1024) ? 2 : 1;
    static constexpr int O_W1 = SPLIT ? SP_NBUF * 2 * SP_OPER_F : 4 * OPER;      // [2 buffers][gz, h], then the first-layer weights
    static constexpr int SMEM_FLOATS = O_W1 + HP * PINN_XS_LD;
    static constexpr int WGS_PER_CU = (!SPLIT && SMEM_FLOATS * 4 <= 75 * 1024 && AM * BN * 4 <= 64) ? 2 : 1;
};

// SKIPS: the partner of the VAR 8 | 1024 | 128 tile kernels (skip connections 'R ... +' over Tanh / Sigmoid layers): where a skip
// joins BEHIND activation a - 1, h_{a-1} = act(z) + the carried activations, which the tile kernel left in the skip's slab slot
// (one more streamed operand for that layer; a '+' in front of the activation is already part of the saved jets)
// HEAVY: the partner of the full breadth kernels (VAR 8 | 128): any activation of the library per layer (h rebuilt from the value or
// from the pre-activation, whichever the tile kernel saved: pinn_act_saved), skips that carry pre-activation jets included (the slot
// holds whatever was carried)
// ALLACT (round 5): HEAVY with all sixteen activation codes (the partner of the ACTC -2 tile kernels); HEAVY alone knows codes 0 .. 7
template <int HP, int ND, int N2, bool COMB, int MT, bool SPLIT = false, bool SKIPS = false, bool HEAVY = false, bool ALLACT = false>
PINN_GLOBAL void PINN_LAUNCH_BOUNDS2((PinnWgCfg<HP, ND, N2, MT, SPLIT>::NTHREADS), (2 * PinnWgCfg<HP, ND, N2, MT, SPLIT>::WGS_PER_CU))
pinn_wgrad_kernel(const PinnKArgs A) {
    using W = PinnWgCfg<HP, ND, N2, MT, SPLIT>;
    using C = typename W::C;
    constexpr int N2n = pinn_n2(N2), N3n = pinn_n3(N2), N4n = pinn_n4(N2);     // (N2 is the packed count, pinn_kernel.h)
    constexpr bool KEEP0 = (HEAVY || N4n > 0) && N3n > 0;          // keep what was saved of the value stream (d3 / d4 of any activation)
    constexpr int S = W::S, NTW = W::NTW, NTHREADS = W::NTHREADS, LDK = W::LDK, AM = W::AM, BN = W::BN, OPER = W::OPER;
    const int tid = PINN_TID, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int lh = A.lh;
    const float* cw = A.comb_w;
    PINN_SMEM(smem);
    float* W1s = smem + W::O_W1;
    for (int i = tid; i < HP * PINN_XS_LD; i += NTHREADS) {
        const int n = i / PINN_XS_LD, c = i % PINN_XS_LD;
        W1s[i] = (c < A.d) ? A.params[n * A.d + c] : 0.0f;
    }
    float* part = A.partials + (size_t)PINN_BID * A.p_core;
    if (PINN_BID >= A.partial_row0) {
        // this row belongs to no workgroup of the tile kernel (two of these workgroups share a CU at width 128): everything
        // outside the hidden->hidden matrices is zero
        for (int i = tid; i < A.p_core; i += NTHREADS) {
            const int rel = i - A.off_wh;
            const bool in_w = rel >= 0 && rel < lh * A.hidden_stride && (rel % A.hidden_stride) < HP * HP;
            if (!in_w) part[i] = 0.0f;
        }
    }
    PINN_SYNC();
    const int m0 = (wave / W::WN) * AM * 16, n0 = (wave % W::WN) * BN * 16;
    const size_t sv_tile = C::slab_vec4_per_wg(lh, SKIPS ? 2 * A.n_skips : 0), gz_tile = C::gz_vec4_per_tile(lh);
    const long long t_first = A.tile_begin + PINN_BID, t_step = PINN_NBLK;
    auto unit0 = [&](int j) { return (wave * NTW + j) * 16 + 4 * lq; };

    constexpr int NBLK = W::NBLK, HB = W::HB;
    // (width 512: waves [0, NW / 2) hold the units of block 0 in the tile kernel's lane-private layout, the others those of block 1)
    const int my_blk = (NBLK > 1) ? (wave * NTW * 16) / HB : 0;
    for (int li = 0; li < lh; ++li)
    for (int mb = 0; mb < NBLK; ++mb)
    for (int nb = 0; nb < NBLK; ++nb) {
        const bool need_g = (NBLK == 1) || my_blk == mb, need_h = (NBLK == 1) || my_blk == nb;
        // layer a = li + 1: A operand gz_a, B operand h_{a-1} = h of activation index li
        const PinnAct act(pinn_act_code(A.act_codes, li) & (HEAVY ? (ALLACT ? 15 : 7) : 1), HEAVY ? A.act_par[li] : 0.0f);
        int sk_in = -1;                   // skip that joins behind activation li
        if (SKIPS) for (int i = 0; i < A.n_skips; ++i) if (A.skip_dst[i] == li && !((A.skip_pre >> i) & 1)) sk_in = i;
        f32x4 acc[AM][BN];
#pragma unroll
        for (int i = 0; i < AM; ++i)
#pragma unroll
            for (int jn = 0; jn < BN; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

        // (the lambdas below are forced inline: at width 512 -- four unit tiles per lane -- hipcc's inliner left load_raw / transform as
        //  FUNCTIONS, their array arguments and the by-value kernel arguments in scratch: 12 ms instead of 2.6 for 65 536 points, round 6)
        // raw operands of a stage, straight from HBM (the tile kernel's lane-private layout: this thread reads what the
        // thread with the same id stored)
        auto load_raw = [&](long long tile, int mt, int s, f32x4 (&gzr)[NTW], f32x4 (&svr)[NTW], f32x4 (&skr)[SKIPS ? NTW : 1]) PINN_INLINE_LAMBDA {
            // (uniform 64-bit base per slot + this thread's 32-bit index: scalar address arithmetic, one VGPR of offset)
            // (debug flag 2, timing experiments only: every stage re-reads the first tile -- operands from L2 instead of HBM)
            const size_t tl = PINN_DBG(A, 2) ? 0 : (size_t)(tile - A.tile_begin);
            const f32x4* gzp = A.gzslab + tl * gz_tile + (((size_t)(li * S + s) * NTW) * MT + mt) * NTHREADS;
            const f32x4* svp = A.slab + tl * sv_tile + (((size_t)(li * S + s) * NTW) * MT + mt) * NTHREADS;
            const unsigned t = (unsigned)tid;
            if (SKIPS && sk_in >= 0 && need_h) {
                const f32x4* skp = A.slab + tl * sv_tile + (((size_t)((lh + 1 + sk_in) * S + s) * NTW) * MT + mt) * NTHREADS;
#pragma unroll
                for (int j = 0; j < NTW; ++j) skr[SKIPS ? j : 0] = pinn_ld4_stream<(HP >= PINN_SLAB_NT_MIN_HP)>(skp + (size_t)j * MT * NTHREADS + t);
            }
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                if (need_g) gzr[j] = pinn_ld4_stream<(HP >= PINN_SLAB_NT_MIN_HP)>(gzp + (size_t)j * MT * NTHREADS + t);
                if (!need_h) continue;
                if (li > 0 || s == 0) {
                    svr[j] = pinn_ld4_stream<(HP >= PINN_SLAB_NT_MIN_HP)>(svp + (size_t)j * MT * NTHREADS + t);
                } else {
                    // first layer: only tanh(z) was saved; z_k = W1[:, col_k] (+ the diagonal partner), z_kk = 0
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        svr[j][r] = (s <= ND) ? pinn_dir_weight<pinn_dir_x(N2)>(W1s + (unit0(j) + r) * PINN_XS_LD, A.dir_cols[s - 1 < 0 ? 0 : s - 1]) : 0.0f;
                }
            }
        };
        // h_s of the saved jets, stream by stream in ascending s (d1, d2 and the z_k^2 terms ride along in registers: one
        // running sum_k c_k z_k^2 for the combined second-order stream, else z_k^2 of the N2 directions that have one)
        f32x4 d1v[NTW], d2v[NTW], zz[(COMB || N2n == 0) ? 1 : N2n][NTW];
        f32x4 z1v[N3n > 0 ? N3n : 1][NTW], z2v[N3n > 0 ? N3n : 1][NTW];      // first / second streams of the third-order directions
        f32x4 s0v[KEEP0 ? NTW : 1];                                           // what was saved of the value stream (third / fourth derivative of the activation)
        f32x4 z3v[N4n > 0 ? N4n : 1][NTW];                                    // third streams of the fourth-order directions
        auto transform = [&](int s, const f32x4 (&svr)[NTW], f32x4 (&hv)[NTW], const f32x4 (&skr)[SKIPS ? NTW : 1]) PINN_INLINE_LAMBDA {
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (s == 0) {
                        float d1, d2;
                        pinn_act_d12(svr[j][r], act, d1, d2);
                        d1v[j][r] = d1; d2v[j][r] = d2;
                        hv[j][r] = pinn_act_value(svr[j][r], act);
                        if (KEEP0) s0v[KEEP0 ? j : 0][r] = svr[j][r];
                        if (COMB) zz[0][j][r] = 0.0f;
                    } else if (s <= ND) {
                        hv[j][r] = d1v[j][r] * svr[j][r];
                        if (COMB) zz[0][j][r] = fmaf(cw[s - 1], svr[j][r] * svr[j][r], zz[0][j][r]);
                        else if (s - 1 < N2n) zz[(COMB || N2n == 0) ? 0 : s - 1][j][r] = svr[j][r] * svr[j][r];
                        if (s - 1 < N3n) z1v[N3n > 0 ? s - 1 : 0][j][r] = svr[j][r];
                    } else if (s <= ND + N2n) {
                        const float q = zz[(COMB || N2n == 0) ? 0 : s - 1 - ND][j][r];
                        hv[j][r] = fmaf(d2v[j][r], q, d1v[j][r] * svr[j][r]);
                        if (s - 1 - ND < N3n) z2v[N3n > 0 ? s - 1 - ND : 0][j][r] = svr[j][r];
                    } else if (s <= ND + N2n + N3n) {
                        // third order: h3 = a1 z3 + 3 a2 z1 z2 + a3 z1^3 with the activation's derivatives a1, a2, a3; a3 from
                        // a1 (tanh: a1 (4 - 6 a1), sigmoid: a1 (1 - 6 a1); HEAVY: any activation, from what was saved of the value stream)
                        const int k = N3n > 0 ? s - 1 - ND - N2n : 0;
                        const float d1 = d1v[j][r], d2 = d2v[j][r], z1 = z1v[k][j][r], z2 = z2v[k][j][r];
                        const float d3 = KEEP0 ? pinn_act_d3(s0v[KEEP0 ? j : 0][r], d1, d2, act)
                                               : (act == PINN_ACT_TANH) ? d1 * (4.0f - 6.0f * d1) : d1 * (1.0f - 6.0f * d1);
                        hv[j][r] = d1 * svr[j][r] + 3.0f * d2 * z1 * z2 + d3 * z1 * z1 * z1;
                        if (N4n > 0 && k < N4n) z3v[N4n > 0 ? (k < N4n ? k : 0) : 0][j][r] = svr[j][r];
                    } else {
                        // fourth order (round 5): h4 = a1 z4 + a2 (4 z1 z3 + 3 z2^2) + 6 a3 z1^2 z2 + a4 z1^4
                        const int k = N4n > 0 ? s - 1 - ND - N2n - N3n : 0;
                        const float d1 = d1v[j][r], d2 = d2v[j][r], z1 = z1v[k][j][r], z2 = z2v[k][j][r], z3 = z3v[k][j][r];
                        const float sv0 = s0v[KEEP0 ? j : 0][r];
                        const float d3 = pinn_act_d3(sv0, d1, d2, act), d4 = pinn_act_d4(sv0, d1, d2, act);
                        hv[j][r] = d1 * svr[j][r] + d2 * (4.0f * z1 * z3 + 3.0f * z2 * z2) + 6.0f * d3 * z1 * z1 * z2 + d4 * z1 * z1 * z1 * z1;
                    }
                }
                if (SKIPS && sk_in >= 0) hv[j] += skr[SKIPS ? j : 0];
            }
        };
        auto write_stage = [&](float* buf, const f32x4 (&gzr)[NTW], const f32x4 (&hv)[NTW]) PINN_INLINE_LAMBDA {
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = unit0(j) + r - my_blk * HB;          // (this lane's unit inside its block)
                    if (need_g) buf[row * LDK + lr] = gzr[j][r];
                    if (need_h) buf[OPER + row * LDK + lr] = hv[j][r];
                }
        };

        // stages of this workgroup in order: (tile, mt, s), s fastest and unrolled (the jet formulas branch on s). Stage i
        // computes from LDS buffer p while the raw operands of stage i + 1 are in flight from HBM; they are turned into
        // buffer p ^ 1 right behind the MFMAs, one barrier per stage.
        auto mfma_stage = [&](const float* bg) PINN_INLINE_LAMBDA {
            // B fragments of the whole stage up front; A fragment of output tile row i + 1 in flight while the 4 * BN MFMAs
            // of row i issue (pinned with sched barriers: left alone the scheduler hoists every fragment load to the top
            // and the 128 accumulators no longer fit beside them)
            const float* bh = bg + OPER;
            f32x4 bf[BN];
#pragma unroll
            for (int jn = 0; jn < BN; ++jn) bf[jn] = pinn_ld4(bh + (n0 + 16 * jn + lr) * LDK + 4 * lq);
            f32x4 af = pinn_ld4(bg + (m0 + lr) * LDK + 4 * lq);
#pragma unroll
            for (int i = 0; i < AM; ++i) {
                PINN_SCHED_BARRIER();
                f32x4 afn = af;
                if (i + 1 < AM) afn = pinn_ld4(bg + (m0 + 16 * (i + 1) + lr) * LDK + 4 * lq);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int jn = 0; jn < BN; ++jn) acc[i][jn] = pinn_mfma16(af[kk], bf[jn][kk], acc[i][jn]);
                if (i + 1 < AM) pinn_sched_reads_first<1, 4 * BN>();
                PINN_SCHED_BARRIER();
                af = afn;
            }
        };
        if constexpr (SPLIT) {
            // ---- split-bf16 form: stages in PAIRS (K = 32), operands split exactly into hi + mid + lo bf16 (pinn_split3), six
            //      partial products per MFMA step on v_mfma_f32_16x16x32_bf16 (pinn_kernel.h, VAR 512 of the tile kernel) ----
            long long n_tiles_wg = 0;
            if (t_first < A.tile_end) n_tiles_wg = (A.tile_end - t_first + t_step - 1) / t_step;
            const long long n_stages = n_tiles_wg * MT * S;
            constexpr int GROUP = (S % 2 == 0) ? S : 2 * S;          // stages per unrolled group (even; the stream index of every stage a compile-time fact)
            constexpr int NB = W::SP_NBUF;
            char* lds = reinterpret_cast<char*>(smem);
            auto buf_of = [&](int b) { return lds + (size_t)b * 2 * W::SP_OPER_F * 4; };
            // dword of (unit row, k pair lr) inside a plane
            auto sp_dword = [&](int row, int kd) { return row * 64 + ((((kd >> 2) ^ (3 * ((row >> 2) & 1))) & 3) << 4) + (kd & 3) * 4; };
            f32x4 gz0[NTW], h0[NTW];                                 // first stage of the pair under construction
            auto write_pair = [&](char* buf, const f32x4 (&g0)[NTW], const f32x4 (&hh0)[NTW], const f32x4 (&g1)[NTW], const f32x4 (&hh1)[NTW]) {
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        unsigned a0[3], a1[3], b0[3], b1[3];
                        pinn_split3(g0[j][r], a0[0], a0[1], a0[2]);
                        pinn_split3(g1[j][r], a1[0], a1[1], a1[2]);
                        pinn_split3(hh0[j][r], b0[0], b0[1], b0[2]);
                        pinn_split3(hh1[j][r], b1[0], b1[1], b1[2]);
                        const int off = sp_dword(unit0(j) + r, lr);
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            *reinterpret_cast<unsigned*>(buf + p * W::SP_PLANE_F * 4 + off) = pinn_pack_hi16(a0[p], a1[p]);
                            *reinterpret_cast<unsigned*>(buf + (W::SP_OPER_F + p * W::SP_PLANE_F) * 4 + off) = pinn_pack_hi16(b0[p], b1[p]);
                        }
                    }
            };
            auto frag = [&](const char* oper, int row, pinn_s16x8 (&f)[3]) {
                const int off = row * 64 + (((lq ^ (3 * ((row >> 2) & 1))) & 3) << 4);
#pragma unroll
                for (int p = 0; p < 3; ++p) f[p] = *reinterpret_cast<const pinn_s16x8*>(oper + p * W::SP_PLANE_F * 4 + off);
            };
            auto mfma_pair = [&](const char* buf) {
                constexpr int pa[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, pb[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
                constexpr int BH = (BN > 2) ? 2 : BN;               // B fragments two output columns at a time (registers)
                const char* bh = buf + W::SP_OPER_F * 4;
#pragma unroll
                for (int jb = 0; jb < BN; jb += BH) {
                    pinn_s16x8 bf[BH][3], af[3];
#pragma unroll
                    for (int jn = 0; jn < BH; ++jn) frag(bh, n0 + 16 * (jb + jn) + lr, bf[jn]);
                    frag(buf, m0 + lr, af);
#pragma unroll
                    for (int i = 0; i < AM; ++i) {
                        PINN_SCHED_BARRIER();
                        pinn_s16x8 afn[3];
                        if (i + 1 < AM) frag(buf, m0 + 16 * (i + 1) + lr, afn);
#pragma unroll
                        for (int t = 9 - PINN_SP_NPROD; t < 9; ++t)
#pragma unroll
                            for (int jn = 0; jn < BH; ++jn) acc[i][jb + jn] = pinn_mfma16_bf16(af[pa[t]], bf[jn][pb[t]], acc[i][jb + jn]);
                        if (i + 1 < AM) pinn_sched_reads_first<3, PINN_SP_NPROD * BH>();
                        PINN_SCHED_BARRIER();
                        if (i + 1 < AM) {
#pragma unroll
                            for (int p = 0; p < 3; ++p) af[p] = afn[p];
                        }
                    }
                }
            };
            auto stage_pos = [&](long long i, long long& tile, int& mt) {
                const long long g = i / S;
                tile = t_first + (g / MT) * t_step;
                mt = (int)(g % MT);
            };
            // raw operands of the NEXT pair are requested from HBM before the MFMAs of the current one
            f32x4 gzr[2][NTW], svr[2][NTW], skr[2][SKIPS ? NTW : 1];
            auto load_pair = [&](long long i0, int s_a, int s_b) {
                long long tile; int mt;
                if (i0 < n_stages) { stage_pos(i0, tile, mt); load_raw(tile, mt, s_a, gzr[0], svr[0], skr[0]); }
                if (i0 + 1 < n_stages) { stage_pos(i0 + 1, tile, mt); load_raw(tile, mt, s_b, gzr[1], svr[1], skr[1]); }
            };
            auto build_pair = [&](char* buf, long long i0, int s_a, int s_b) {
                f32x4 ha[NTW], hb[NTW], ga[NTW], gb[NTW];
#pragma unroll
                for (int j = 0; j < NTW; ++j) { ga[j] = gzr[0][j]; gb[j] = f32x4{0.f, 0.f, 0.f, 0.f}; hb[j] = f32x4{0.f, 0.f, 0.f, 0.f}; ha[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                if (i0 < n_stages) transform(s_a, svr[0], ha, skr[0]);
                else {
#pragma unroll
                    for (int j = 0; j < NTW; ++j) ga[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (i0 + 1 < n_stages) {
                    transform(s_b, svr[1], hb, skr[1]);
#pragma unroll
                    for (int j = 0; j < NTW; ++j) gb[j] = gzr[1][j];
                }
                write_pair(buf, ga, ha, gb, hb);
            };
            int pbuf = 0;
            if (n_stages > 0) {
                load_pair(0, 0, 1 % S);
                build_pair(buf_of(0), 0, 0, 1 % S);
            }
            PINN_SYNC();
            for (long long g0 = 0; g0 < n_stages; g0 += GROUP) {
#pragma unroll
                for (int u = 0; u < GROUP; u += 2) {
                    const long long i = g0 + u;                      // first stage of this pair
                    if (i < n_stages) {
                        const int s_na = (u + 2) % S, s_nb = (u + 3) % S;       // streams of the next pair (compile time)
                        const bool has_next = i + 2 < n_stages;
                        if (has_next) load_pair(i + 2, s_na, s_nb);
                        mfma_pair(buf_of(pbuf));
                        if (NB == 1) PINN_SYNC();                     // one buffer: everybody is done reading it
                        if (has_next) build_pair(buf_of(NB == 2 ? pbuf ^ 1 : 0), i + 2, s_na, s_nb);
                        PINN_SYNC();
                        if (NB == 2) pbuf ^= 1;
                    }
                }
            }
        } else {
        // (the HBM loads of stage i + 1 run during the MFMAs of stage i, one register set. Two sets with the loads two stages ahead
        //  measured 1-5 % SLOWER on MI355X: the loads are not what the waves wait for -- DESIGN.md section 6a)
        f32x4 gzr[NTW], svr[NTW], hv[NTW], skr[SKIPS ? NTW : 1];
        int p = 0;
        if (t_first < A.tile_end) {
            load_raw(t_first, 0, 0, gzr, svr, skr);
            if (need_h) transform(0, svr, hv, skr);
            write_stage(smem, gzr, hv);
        }
        PINN_SYNC();
        for (long long tile = t_first; tile < A.tile_end; tile += t_step) {
            for (int mt = 0; mt < MT; ++mt) {
                const bool last_group = (mt + 1 == MT) && (tile + t_step >= A.tile_end);
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    const int ns = (k + 1) % S;                          // stream of the next stage (compile time)
                    const bool has_next = !(last_group && k + 1 == S);
                    const int nmt = (k + 1 == S) ? (mt + 1) % MT : mt;
                    const long long ntile = (k + 1 == S && mt + 1 == MT) ? tile + t_step : tile;
                    // (debug flags 8 / 16 / 32, timing experiments only: no barrier / no LDS staging / no HBM loads)
                    if (has_next && !PINN_DBG(A, 32)) load_raw(ntile, nmt, ns, gzr, svr, skr);
                    mfma_stage(smem + p * 2 * OPER);
                    if (has_next && !PINN_DBG(A, 16)) {
                        if (need_h) transform(ns, svr, hv, skr);
                        write_stage(smem + (p ^ 1) * 2 * OPER, gzr, hv);
                    }
                    if (!PINN_DBG(A, 8)) PINN_SYNC();
                    p ^= 1;
                }
            }
        }
        }
        // this workgroup's dW_li: D[(l >> 4) * 4 + r][l & 15] of tile (i, jn) -> row (out) m0 + 16 i + 4 lq + r, column (in) n0 + 16 jn + lr
        // (one 32-bit lane offset + a pointer per output tile row i; rows r and column tiles jn sit at immediate offsets below 4 KB: left
        //  to itself the compiler hoists all AM * BN * 4 sixty-four-bit element offsets out of the layer loop and parks them in scratch
        //  -- 200 spilled registers at width 256, none of them inside a stage, but 800 B of scratch per lane for nothing)
        if (NBLK > 1) PINN_SYNC();          // (the next block pass restages both LDS buffers from their start)
        float* dst = part + A.off_wh + (size_t)li * A.hidden_stride + (unsigned)((mb * HB + m0 + 4 * lq) * HP + nb * HB + n0 + lr);
#pragma unroll
        for (int i = 0; i < AM; ++i) {
            float* row = dst + (unsigned)(16 * i * HP);
#pragma unroll
            for (int jn = 0; jn < BN; ++jn)
#pragma unroll
                for (int r = 0; r < 4; ++r) row[r * HP + 16 * jn] = acc[i][jn][r];
        }
    }
}
