// pinn_abi.cpp -- C-ABI of libpinn_hip.so (declared in include/pinn.h): descriptor, layout, launch dispatch.
// Built with hipcc for gfx950 (product) or, with -DPINN_EMU, as tests/emu/_build/libpinn_emu.so (test only).
#include "pinn_inst.h"
#include "pinn_fit_kernel.h"
#include "pinn_aux_kernels.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

namespace {
thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

int round16(int v) { return (v + 15) / 16 * 16; }

int g_profile = 0;
long long* g_phase_prof = nullptr;
#ifndef PINN_EMU
// measurement hook (pinn_profile_tile): one pair of events per kernel and DEVICE -- an event belongs to the device it was
// created on, and a process may drive several
struct ProfEvents {
    hipEvent_t tile0 = nullptr, tile1 = nullptr, wg0 = nullptr, wg1 = nullptr;
    bool have_tile = false, have_wgrad = false;
};
ProfEvents g_prof[64];
int g_prof_dev = 0;                 // device of the last bracket (what pinn_last_*_ms reads)
ProfEvents* prof_events() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    ProfEvents& e = g_prof[dev];
    if (!e.tile0 && (hipEventCreate(&e.tile0) != hipSuccess || hipEventCreate(&e.tile1) != hipSuccess ||
                     hipEventCreate(&e.wg0) != hipSuccess || hipEventCreate(&e.wg1) != hipSuccess)) return nullptr;
    g_prof_dev = dev;
    return &e;
}
float prof_elapsed(hipEvent_t a, hipEvent_t b) {
    float ms = -1.0f;
    if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess) return -1.0f;
    return ms;
}
#endif
}  // namespace

// (shared with the per-width translation units, not exported)
#define PINN_HIDDEN __attribute__((visibility("hidden")))
PINN_HIDDEN int g_pinn_last_kernel = -1;
PINN_HIDDEN char g_pinn_last_kernel_name[96] = "";
PINN_HIDDEN char g_pinn_last_wgrad_name[96] = "";
PINN_HIDDEN int g_pinn_last_launch[4] = {0, 0, 0, 0};   // grid, workgroups per CU of the plan, threads, dynamic LDS bytes
namespace {
// set while pinn_fit_steps_graph captures a chunk: the sample and reduce launches then read their per-iteration values from the device
// control block (PinnFitCtrl) at index `k` instead of taking them by value
struct FitCapture { const PinnFitCtrl* ctrl; int k; };
thread_local FitCapture g_fit_capture = {nullptr, 0};
// set by pinn_fit_steps around the step of iteration k < K - 1: its reduction launch also draws the batch of iteration k + 1
thread_local PinnNextBatch g_fit_next = {nullptr, 0, {}, 0u, 0u, 0ull};
// set by pinn_fit_steps_graph around ONE residual step call: run_train then launches the one-launch fit chunk (pinn_fit_kernel) over the
// tiles of that step instead of tile kernel + reduction; `done` says whether it did (0: the plan does not qualify -- nothing was launched)
struct FitPersist { PinnFitP p; int active, done; };
thread_local FitPersist g_fit_persist = {{}, 0, 0};
int g_pinn_debug_flags = 0;         // -DPINN_DEBUG_ABI builds: pinn_debug_set_flags
}

struct pinn_net {
    pinn_layout_t lay;
    int n_layers, act, ndims, nparams, has_bc, has_ic, nsp;      // act: uniform activation code or -1
    unsigned long long act_codes[2];                              // 4 bits per activation index (pinn_act_code)
    float act_par[PINN_MAX_LAYERS];                               // parameter of activation a (pinn_set_act_params; torch's defaults at creation)
    int n_skips, skip_src[PINN_MAX_SKIPS], skip_dst[PINN_MAX_SKIPS], skip_pre, skip_src_pre, skip_outer;
    int dims[PINN_MAX_LAYERS + 1];
    float lo[PINN_MAX_INPUTS], hi[PINN_MAX_INPUTS], bc_value;
    int n_cu;
    int gemm_mode;                  // PINN_GEMM_FP32 / PINN_GEMM_BF16X3 (pinn_set_gemm_mode)
    int tanh_mode;                  // PINN_TANH_FAST / PINN_TANH_ACCURATE (pinn_set_tanh_mode)
    // pinn_fit_steps_graph: the instantiated launch graph of one chunk and the arguments it was captured with (host state of the
    // descriptor; no device memory: the control block is the caller's)
    void* fit_graph_exec;
    unsigned long long fit_graph_key[4];
    int fit_graph_k;
    // diagnostics of THIS descriptor (pinn_debug_*: tests run one net at 1 workgroup per CU, with a separate pre-pass launch, with a
    // tiny slab budget -- a second Solver in the same process keeps its own planning)
    int fit_persistent;             // pinn_debug_fit_persistent: small fit chunks as ONE launch. 2 (default): the one-CU form for batches of a few tiles;
                                    // 1: the grid form (measured slower than launch-graph replay on MI355X); 0: never
    int fit_onecu_rounds;           // one-CU form: at most this many sweeps of its virtual workgroups per iteration (pinn_debug_fit_onecu_rounds)
    int max_per_cu;                 // pinn_debug_max_wgs_per_cu (default 4)
    int prepass_in_kernel;          // pinn_debug_prepass_in_kernel (default 1)
    size_t wgx_chunk_bytes;         // pinn_debug_wgx_chunk_bytes (default PINN_WGX_CHUNK_DEFAULT)
};

namespace {
typedef int (*launch_fn)(int, int, const PinnKArgs*, int, void*, int, long long*);
typedef int (*wgrad_fn)(int, int, int, int, const PinnKArgs*, int, void*, int, long long*);

wgrad_fn wgrad_launcher_for(int hp) {
    switch (hp) {
        case 128: return pinn_launch_wgrad_hp128;
        case 256: return pinn_launch_wgrad_hp256;
        case 512: return pinn_launch_wgrad_hp512;
        default: return nullptr;
    }
}

// WGX kernels keep the saved jets and gz of EVERY tile of a launch in HBM (44 B per point, layer and unit at S = 4): a
// batch larger than this many bytes of slab goes through the kernels chunk by chunk (gradients accumulate in the reduction)
constexpr size_t PINN_WGX_CHUNK_DEFAULT = (size_t)6656 << 20;

launch_fn launcher_for(int hp) {
    switch (hp) {
        case 16: return pinn_launch_tile_hp16;
        case 32: return pinn_launch_tile_hp32;
        case 64: return pinn_launch_tile_hp64;
        case 128: return pinn_launch_tile_hp128;
        case 256: return pinn_launch_tile_hp256;
        case 512: return pinn_launch_tile_hp512;
        default: return nullptr;
    }
}

// smallest compiled (nd, n2k) with n2k >= n2 (extra second-derivative streams get zero upstream gradient)
int pick_n2(int nd, int n2) {
    // (nd = 4: first derivatives only; its second-order form exists as ONE combined stream, `comb`)
    static const int avail[5][5] = {{1, 0, 0, 0, 0}, {1, 1, 0, 0, 0}, {1, 0, 1, 0, 0}, {0, 0, 1, 1, 0}, {1, 0, 0, 0, 0}};
    if (pinn_n3(n2) > 0 || pinn_n4(n2) > 0)     // packed count with third- / fourth-order streams: exact instantiations only
        return (((nd == 1 || nd == 2) && n2 == 9) || (nd == 1 && n2 == 73)) ? n2 : -1;       // 73 = 1 | 1 << 3 | 1 << 6: u, u', u'', u''', u''''
    if (n2 > 7) return -1;
    if (nd < 0 || nd > 4 || n2 < 0 || n2 > nd) return -1;
    for (int k = n2; k <= nd; ++k)
        if (avail[nd][k]) return k;
    return -1;
}

// compute units of the device the CALL runs on (the current device of the calling thread; cached per device): a net
// created while another device was current still gets the right grid
int device_cus(const pinn_net* net) {
#ifndef PINN_EMU
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return net->n_cu;
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : -1;
    }
    return cus[dev] > 0 ? cus[dev] : net->n_cu;
#else
    return net->n_cu;
#endif
}

struct Plan {
    launch_fn fn;
    int n2k, grid, threads, per_cu;
    size_t smem, slab_vec4_per_wg;  // saved-jet slab per workgroup (per TILE for WGX kernels)
    size_t wt_floats_per_wg;        // slab-in-LDS kernels: W^T scratch of every workgroup (PinnKArgs::wt)
    int64_t ntiles;                 // tiles of T = 16 * mt points in the batch
    int mt, comb;
    // WGX (widths >= 128): hidden->hidden weight gradients by pinn_wgrad_kernel from per-tile slabs in HBM
    int wgx, grid2;                 // grid2: workgroups of the weight-gradient kernel
    int wt_global;                  // the kernel reads the transposed global copy of the hidden weights (pinn_transpose_kernel)
    int split;                      // split-bf16 kernel: reads the bf16 fragment copy of the hidden weights (pinn_wsplit_kernel)
    int prepass_floats_per_lane;    // LDS floats per point of a tile the in-kernel pre-pass may use for its (double) registers; 0: no in-kernel pre-pass
    wgrad_fn wfn;
    size_t gz_vec4_per_tile;
    int64_t chunk_tiles;            // tiles per pass through the two kernels
    int rows() const { return wgx && grid2 > grid ? grid2 : grid; }       // partial-gradient rows the reduction sums
    size_t slab_bytes() const { return (size_t)(wgx ? chunk_tiles : grid) * slab_vec4_per_wg * 16; }
    size_t gz_bytes() const { return wgx ? (size_t)chunk_tiles * gz_vec4_per_tile * 16 : 0; }
};

// `hint`: the arguments of the call being planned. The launcher picks shape-specialised instantiations from them
// (pinn_spec_of), and those may differ from the general kernel in everything a plan holds -- workgroups per CU, tile
// height, slab size, whether the transposed weight copy is needed -- so a call is planned with its own arguments;
// sizing queries (no call yet) plan with the general probe AND with the arguments a typical training step of this net
// would carry (typical_step_args) and take the larger answer.
int make_plan(const pinn_net* net, int64_t n_points, int nd, int n2, Plan* plan, int mode = PINN_MODE_FORWARD,
              int res_kind = PINN_RES_PROGRAM, int comb = 0, const PinnKArgs* hint = nullptr) {
    plan->fn = launcher_for(net->lay.hp);
    if (!plan->fn) return fail("no kernel for padded hidden width %d (supported: 16, 32, 64, 128, 256, 512)", net->lay.hp);
    plan->n2k = comb ? 1 : pick_n2(nd, n2);
    if (comb && (n2 != 1 || nd < 2 || nd > 4)) return fail("combined second-order stream needs n2 == 1 and nd in {2, 3, 4}");
    if (plan->n2k < 0) return fail("unsupported derivative spec nd=%d n2=%d n3=%d n4=%d (nd <= 3 with n2 <= nd, nd = 4 with n2 = 0 / one combined second-order stream, one third-order direction with nd <= 2, or one fourth-order direction alone)", nd, pinn_n2(n2), pinn_n3(n2), pinn_n4(n2));
    PinnKArgs probe;
    if (hint) {
        probe = *hint;
    } else {
        memset(&probe, 0, sizeof(probe));
        probe.lh = net->lay.lh;
        probe.act = net->act;
        probe.act_codes[0] = net->act_codes[0]; probe.act_codes[1] = net->act_codes[1];
        for (int i = 0; i < PINN_MAX_LAYERS; ++i) probe.act_par[i] = net->act_par[i];
        probe.n_skips = net->n_skips;
        probe.skip_pre = net->skip_pre;
        probe.skip_src_pre = net->skip_src_pre;
        probe.skip_outer = net->skip_outer;
    }
    probe.gemm_mode = net->gemm_mode;
    probe.tanh_mode = net->tanh_mode;
    probe.mode = mode;
    probe.res_kind = res_kind;
    probe.comb = comb;
    long long info[16] = {0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0};
    if (plan->fn(nd, plan->n2k, &probe, 0, nullptr, 1, info))
        return fail("no kernel instantiation for width %d with nd=%d n2=%d%s", net->lay.hp, nd, plan->n2k, comb ? " (combined)" : "");
    plan->smem = (size_t)info[0];
    plan->threads = (int)info[1];
    plan->slab_vec4_per_wg = (size_t)info[2];
    plan->wt_floats_per_wg = (size_t)info[5];
    const int64_t ntiles = (n_points + 15) / 16;
    const int64_t wg_tiles = (ntiles + info[4] - 1) / info[4];       // a two-team workgroup streams two tiles at a time
    if (info[3] > net->max_per_cu) info[3] = net->max_per_cu;
    plan->per_cu = (int)info[3];
    int64_t grid = (int64_t)device_cus(net) * info[3];
    const int64_t teams = info[10] > 0 ? info[10] : 1;              // a two-team workgroup streams two tiles at a time
    if (grid > (wg_tiles + teams - 1) / teams) grid = (wg_tiles + teams - 1) / teams;
    if (grid < 1) grid = 1;
    plan->grid = (int)grid;
    plan->ntiles = wg_tiles;
    plan->mt = (int)info[4];
    plan->comb = comb;
    plan->wgx = (mode != PINN_MODE_FORWARD && info[6]) ? 1 : 0;
    plan->wt_global = (int)info[9];
    plan->split = (int)info[11];
    plan->prepass_floats_per_lane = (int)info[12];
    plan->grid2 = 0; plan->wfn = nullptr; plan->gz_vec4_per_tile = 0; plan->chunk_tiles = wg_tiles;
    if (plan->wgx) {
        plan->wfn = wgrad_launcher_for(net->lay.hp);
        long long winfo[12] = {0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 1, 0};
        if (!plan->wfn || plan->wfn(nd, plan->n2k, comb, plan->mt, &probe, 0, nullptr, 1, winfo))
            return fail("no weight-gradient kernel for width %d with nd=%d n2=%d%s", net->lay.hp, nd, plan->n2k, comb ? " (combined)" : "");
        int64_t grid2 = (int64_t)device_cus(net) * winfo[3];
        if (grid2 > wg_tiles) grid2 = wg_tiles;
        plan->grid2 = (int)(grid2 < 1 ? 1 : grid2);
        // the hidden->hidden blocks of a partial row are written by pinn_wgrad_kernel alone (rows < grid2; the tile kernel leaves
        // them untouched in WGX mode), and the reduction sums max(grid, grid2) rows: a tile-kernel grid beyond grid2 would add
        // rows whose blocks nobody wrote -- the occupancy query may answer more workgroups per CU for the tile
        // kernel than WGS_PER_CU of the weight-gradient kernel, so the tile grid is capped here
        if (plan->grid > plan->grid2) plan->grid = plan->grid2;
        plan->gz_vec4_per_tile = (size_t)info[7];
        // whole sweeps of the persistent workgroups per chunk
        const size_t per_tile = (plan->slab_vec4_per_wg + plan->gz_vec4_per_tile) * 16;
        const int64_t sweep = plan->grid2 > plan->grid ? plan->grid2 : plan->grid;
        int64_t chunk = (int64_t)(net->wgx_chunk_bytes / (per_tile ? per_tile : 1)) / sweep * sweep;
        if (chunk < sweep) chunk = sweep;
        plan->chunk_tiles = chunk < wg_tiles ? chunk : wg_tiles;
    }
    return 0;
}

void note_launch(const Plan& plan) {
    g_pinn_last_launch[0] = plan.grid; g_pinn_last_launch[1] = plan.per_cu;
    g_pinn_last_launch[2] = plan.threads; g_pinn_last_launch[3] = (int)plan.smem;
}

void fill_args(const pinn_net* net, PinnKArgs* a, const float* params, const float* xs, int64_t n, const int* dir_cols,
               int nd, int n2, const float* ic_streams, float ic_const) {
    memset(a, 0, sizeof(*a));
    const pinn_layout_t& L = net->lay;
    a->params = params; a->xs = xs; a->ic_streams = ic_streams; a->n_points = n;
    a->lh = L.lh; a->d = L.d; a->act = net->act; a->act_codes[0] = net->act_codes[0]; a->act_codes[1] = net->act_codes[1];
    for (int i = 0; i < PINN_MAX_LAYERS; ++i) a->act_par[i] = net->act_par[i];
    a->n_skips = net->n_skips;
    a->skip_pre = net->skip_pre;
    a->skip_src_pre = net->skip_src_pre;
    a->skip_outer = net->skip_outer;
    for (int k = 0; k < PINN_MAX_SKIPS; ++k) { a->skip_src[k] = net->skip_src[k]; a->skip_dst[k] = net->skip_dst[k]; }
    a->off_b1 = L.off_b1; a->off_wh = L.off_wh; a->hidden_stride = L.hidden_stride; a->off_wl = L.off_wl;
    a->off_bl = L.off_bl; a->off_ls = L.off_log_scale; a->off_loss = L.off_loss;
    a->p_core = L.p_total;          // partial rows / gradient buffer span the user slots too (V gradients of residual programs)
    a->off_extra = L.off_extra;
    a->ndims = net->ndims; a->nsp = net->nsp; a->has_bc = net->has_bc; a->has_ic = net->has_ic;
    a->bc_value = net->bc_value; a->t0 = net->lo[net->ndims - 1]; a->ic_const = ic_const;
    for (int i = 0; i < PINN_MAX_INPUTS; ++i) {
        a->lo[i] = net->lo[i]; a->hi[i] = net->hi[i];
        a->inv_w[i] = 1.0f / (net->hi[i] - net->lo[i]);
    }
    for (int k = 0; k < PINN_MAX_DIRS; ++k) a->dir_cols[k] = (k < nd) ? dir_cols[k] : 0;
    a->s_user = pinn_ns(nd, n2);
    a->gemm_mode = net->gemm_mode;
    a->tanh_mode = net->tanh_mode;
    a->tile_begin = 0;
    a->tile_end = 0;                // set from the plan (set_tile_range) before every launch
}

void set_tile_range(PinnKArgs* a, const Plan& plan, int64_t begin, int64_t end) {
    a->tile_begin = begin;
    a->tile_end = end < plan.ntiles ? end : plan.ntiles;
}

// what a fused training step of this net most likely passes: direction k = column k, constant affine coefficients
void typical_step_args(const pinn_net* net, PinnKArgs* a, int64_t n, int nd, int n2) {
    int dirs[PINN_MAX_DIRS];
    for (int k = 0; k < PINN_MAX_DIRS; ++k) dirs[k] = k;
    fill_args(net, a, nullptr, nullptr, n, dirs, nd, n2, nullptr, 0.0f);
    for (int s = 0; s < PINN_MAX_STREAMS; ++s) a->coef_row[s] = -1;
    a->src_row = -1;
}

int check_dirs(const pinn_net* net, const int* dir_cols, int nd, int n2p) {
    const int n2 = pinn_n2(n2p), n3 = pinn_n3(n2p), n4 = pinn_n4(n2p);       // packed count: seconds | thirds << 3 | fourths << 6 (include/pinn.h)
    if (nd < 0 || nd > PINN_MAX_DIRS || n2 > nd || n3 > n2 || n4 > n3 || n2p < 0) return fail("bad derivative spec nd=%d n2=%d n3=%d n4=%d", nd, n2, n3, n4);
    for (int k = 0; k < nd; ++k) {
        // direction code: column a, or the diagonal e_a +- e_b as a | (b + 1) << 4 | PINN_DIR_MINUS, | PINN_DIR_DOUBLE: 2 e_a +- e_b (include/pinn.h)
        if (!dir_cols || dir_cols[k] < 0 || dir_cols[k] > 0x7fff) return fail("dir_cols[%d] out of range", k);
        const int a = dir_cols[k] & 15, b = ((dir_cols[k] >> 4) & 15) - 1, c = ((dir_cols[k] >> 10) & 15) - 1;
        if (a >= net->lay.d || b >= net->lay.d || c >= net->lay.d || a == b || (c >= 0 && (c == a || c == b)))
            return fail("dir_cols[%d] names a column outside the %d inputs (or one column twice)", k, net->lay.d);
        if ((dir_cols[k] & (PINN_DIR_MINUS | PINN_DIR_DOUBLE)) && b < 0) return fail("dir_cols[%d]: PINN_DIR_MINUS / PINN_DIR_DOUBLE need a second column", k);
        if (c >= 0 && (b < 0 || (dir_cols[k] & PINN_DIR_DOUBLE))) return fail("dir_cols[%d]: a third column needs a plain second one", k);
        if ((dir_cols[k] & PINN_DIR_MINUS_C) && c < 0) return fail("dir_cols[%d]: PINN_DIR_MINUS_C needs a third column", k);
        if (c >= 0 && k < n4) return fail("dir_cols[%d]: three-column directions carry derivatives up to third order", k);
        // (the kernels decode these extensions only in the stream shapes that can carry them: pinn_dir_x, pinn_kernel.h)
        if (c >= 0 && n3 == 0) return fail("dir_cols[%d]: a three-column direction needs a third-order stream in the call (n3 > 0)", k);
        if ((dir_cols[k] & PINN_DIR_DOUBLE) && n4 == 0) return fail("dir_cols[%d]: PINN_DIR_DOUBLE needs a fourth-order stream in the call (n4 > 0)", k);
        // (round 5: third derivatives along diagonals too -- what mixed third-order partials are assembled from)
    }
    return 0;
}

struct AdamArgs {
    float* params; float* m; float* v; const unsigned char* mask; int* step_ptr;
    int step_value; float lr, b1, b2, eps;
    float* loss_out; int off_loss;
};

int launch_reduce(const float* partials, int n_wg, int p_core, float* grads, int accumulate, void* stream,
                  const AdamArgs* adam = nullptr) {
    const int blocks = (p_core + PINN_REDUCE_PB - 1) / PINN_REDUCE_PB;
    const size_t smem = 1024 * sizeof(double);      // (the chunk sums cross the LDS in double: pinn_reduce_kernel)
    AdamArgs z = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0.f, 0.f, 0.f, nullptr, -1};
    const AdamArgs& a = adam ? *adam : z;
    const int do_adam = adam ? 1 : 0;
    float step_size = 0.0f, bc2_sqrt = 1.0f;
    if (adam) pinn_adam_scalars((double)a.step_value, a.lr, a.b1, a.b2, &step_size, &bc2_sqrt);
#ifdef PINN_EMU
    emu::launch(blocks, 1024, smem, [&] {
        pinn_reduce_kernel(partials, n_wg, p_core, grads, accumulate, do_adam, a.params, a.m, a.v, a.mask, a.step_value,
                           step_size, bc2_sqrt, a.b1, a.b2, a.eps, a.step_ptr, a.loss_out, a.off_loss, nullptr, 0,
                           do_adam ? g_fit_next : PinnNextBatch{nullptr, 0, {}, 0u, 0u, 0ull});
    });
#else
    const PinnFitCtrl* ctrl = do_adam ? g_fit_capture.ctrl : nullptr;
    hipLaunchKernelGGL(pinn_reduce_kernel, dim3(blocks), dim3(1024), smem, (hipStream_t)stream, partials, n_wg, p_core,
                       grads, accumulate, do_adam, a.params, a.m, a.v, a.mask, a.step_value, step_size, bc2_sqrt, a.b1, a.b2,
                       a.eps, a.step_ptr, a.loss_out, a.off_loss, ctrl, g_fit_capture.k,
                       do_adam ? g_fit_next : PinnNextBatch{nullptr, 0, {}, 0u, 0u, 0ull});
    if (hipGetLastError() != hipSuccess) return fail("reduce kernel launch failed");
#endif
    return 0;
}

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// widths >= 128 (PinnCfg::WTG): room for the transposed copy of the lh hidden->hidden matrices
size_t wt_workspace_bytes(const pinn_net* net) {
    return net->lay.lh > 0 ? align256((size_t)net->lay.lh * net->lay.hp * net->lay.hp * sizeof(float)) : 0;
}

// split-bf16 kernels: the bf16 fragment copy of the hidden weights (PinnCfg::wsp_bytes: 2 directions x 3 planes x 2 bytes)
size_t wsp_workspace_bytes(const pinn_net* net) {
    return net->lay.lh > 0 ? align256((size_t)net->lay.lh * 2 * 3 * net->lay.hp * net->lay.hp * 2) : 0;
}
}  // namespace

extern "C" {

const char* pinn_last_error(void) { return g_err; }

int pinn_sample_points(float* xs, int64_t n_points, int d, const int* kind, const float* a, const float* b,
                       uint64_t seed, uint64_t call_index, void* stream) {
    if (!xs || !kind || !a || !b) return fail("null argument");
    if (d < 1 || d > PINN_MAX_INPUTS) return fail("d=%d outside [1, %d]", d, PINN_MAX_INPUTS);
    if (n_points <= 0) return 0;
    PinnSampleSpec spec;
    memset(&spec, 0, sizeof(spec));
    spec.d = d;
    for (int c = 0; c < d; ++c) {
        if (kind[c] < PINN_SAMPLE_UNIFORM || kind[c] > PINN_SAMPLE_CONST) return fail("column %d: unknown sampler kind %d", c, kind[c]);
        spec.kind[c] = kind[c]; spec.a[c] = a[c]; spec.b[c] = b[c];
    }
    const unsigned k0 = (unsigned)(seed & 0xffffffffull), k1 = (unsigned)(seed >> 32);
    const unsigned c_lo = (unsigned)(call_index & 0xffffffffull), c_hi = (unsigned)(call_index >> 32);
    const int blocks = (int)((n_points + 255) / 256);
#ifdef PINN_EMU
    emu::launch(blocks, 256, 0, [&] { pinn_sample_kernel(xs, (long long)n_points, spec, k0, k1, c_lo, c_hi, nullptr, 0); });
#else
    hipLaunchKernelGGL(pinn_sample_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, xs, (long long)n_points, spec,
                       k0, k1, c_lo, c_hi, g_fit_capture.ctrl, g_fit_capture.k);
    if (hipGetLastError() != hipSuccess) return fail("sampler kernel launch failed");
#endif
    return 0;
}

#ifdef PINN_DEBUG_ABI
int pinn_debug_set_flags(int flags) { g_pinn_debug_flags = flags; return 0; }
#endif

int pinn_debug_last_kernel(void) { return g_pinn_last_kernel; }

// torch's default of the one parameter an activation module may carry
static float pinn_act_default_param(int code) {
    return code == PINN_ACT_LEAKYRELU ? 0.01f : (code == PINN_ACT_ELU || code == PINN_ACT_SOFTPLUS) ? 1.0f : 0.0f;
}

int pinn_set_act_params(pinn_t* net, const float* par, int n) {
    if (!net || !par) return fail("null argument");
    if (n != net->n_layers - 1) return fail("pinn_set_act_params: %d values for %d activations", n, net->n_layers - 1);
    for (int a = 0; a < n; ++a) {
        const int code = pinn_act_code(net->act_codes, a);
        const bool takes = code == PINN_ACT_LEAKYRELU || code == PINN_ACT_ELU || code == PINN_ACT_SOFTPLUS;
        if (!takes && par[a] != 0.0f) return fail("pinn_set_act_params: activation %d (code %d) takes no parameter", a, code);
        if (code == PINN_ACT_SOFTPLUS && !(par[a] > 0.0f)) return fail("pinn_set_act_params: Softplus beta must be positive (activation %d: %g)", a, par[a]);
        if (!(par[a] == par[a]) || par[a] > 1e30f || par[a] < -1e30f) return fail("pinn_set_act_params: activation %d: parameter is not finite", a);
    }
    for (int a = 0; a < n; ++a) {
        const int code = pinn_act_code(net->act_codes, a);
        if (code == PINN_ACT_LEAKYRELU || code == PINN_ACT_ELU || code == PINN_ACT_SOFTPLUS) net->act_par[a] = par[a];
    }
    return 0;
}

int pinn_set_tanh_mode(pinn_t* net, int mode) {
    if (!net) return fail("null argument");
    if (mode != PINN_TANH_FAST && mode != PINN_TANH_ACCURATE) return fail("unknown tanh mode %d", mode);
    net->tanh_mode = mode;
    return 0;
}

int pinn_set_gemm_mode(pinn_t* net, int mode) {
    if (!net) return fail("null argument");
    if (mode != PINN_GEMM_FP32 && mode != PINN_GEMM_BF16X3) return fail("unknown GEMM mode %d", mode);
    net->gemm_mode = mode;
    return 0;
}

const char* pinn_last_kernel_name(void) { return g_pinn_last_kernel_name; }

const char* pinn_last_wgrad_kernel_name(void) { return g_pinn_last_wgrad_name; }

int pinn_debug_prepass_in_kernel(pinn_t* net, int enable) {
    if (!net) return fail("null argument");
    net->prepass_in_kernel = enable ? 1 : 0;
    return 0;
}

int pinn_debug_wgx_chunk_bytes(pinn_t* net, long long bytes) {
    if (!net) return fail("null argument");
    net->wgx_chunk_bytes = bytes > 0 ? (size_t)bytes : PINN_WGX_CHUNK_DEFAULT;
    return 0;
}

int pinn_debug_fit_onecu_rounds(pinn_t* net, int rounds) {
    if (!net) return -1;
    const int before = net->fit_onecu_rounds;
    if (rounds >= 1) net->fit_onecu_rounds = rounds;
    return before;
}

// the GRID form of the one-launch fit chunk (pinn_debug_fit_persistent 1) waits device-wide once per iteration with a bounded spin; a
// workgroup that gives up raises a flag and the whole grid returns WITHOUT writing parameters / losses back (pinn_fit_kernel.h). The host
// cannot see that from the launch: this call synchronises with the device and reads the flag of the last such chunk.
static thread_local unsigned* g_fit_last_sync = nullptr;
int pinn_fit_chunk_status(void) {
    if (!g_fit_last_sync) return 0;
#ifdef PINN_EMU
    return (int)g_fit_last_sync[1];
#else
    unsigned host[2] = {0u, 0u};
    if (hipMemcpy(host, g_fit_last_sync, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) { fail("pinn_fit_chunk_status: hipMemcpy failed"); return -1; }
    if (host[1] != 0u) { fail("one-launch fit chunk (grid form) timed out at its device-wide wait: the chunk's iterations were NOT applied"); return 1; }
    return 0;
#endif
}

int pinn_debug_fit_persistent(pinn_t* net, int enable) {
    if (!net) return -1;
    const int before = net->fit_persistent;
    net->fit_persistent = (enable == 1 || enable == 2) ? enable : 0;
    return before;
}

int pinn_debug_max_wgs_per_cu(pinn_t* net, int cap) {
    if (!net) return -1;
    const int before = net->max_per_cu;
    net->max_per_cu = (cap <= 0 || cap > 4) ? 4 : cap;
    return before;
}

int pinn_last_launch_info(int32_t out[4]) {
    if (!out) return fail("null argument");
    for (int i = 0; i < 4; ++i) out[i] = g_pinn_last_launch[i];
    return g_pinn_last_launch[0] > 0 ? 0 : 1;
}

#ifdef PINN_DEBUG_ABI
int pinn_debug_phase_buffer(void* buf) {
    g_phase_prof = reinterpret_cast<long long*>(buf);
    return 0;
}
#endif

int pinn_profile_tile(int enable) {
    g_profile = enable ? 1 : 0;
#ifndef PINN_EMU
    for (ProfEvents& e : g_prof) e.have_tile = e.have_wgrad = false;
#endif
    return 0;
}

float pinn_last_tile_ms(void) {
#ifndef PINN_EMU
    const ProfEvents& e = g_prof[g_prof_dev];
    return e.have_tile ? prof_elapsed(e.tile0, e.tile1) : -1.0f;
#else
    return -1.0f;
#endif
}

float pinn_last_wgrad_ms(void) {
#ifndef PINN_EMU
    const ProfEvents& e = g_prof[g_prof_dev];
    return e.have_wgrad ? prof_elapsed(e.wg0, e.wg1) : -1.0f;
#else
    return -1.0f;
#endif
}

const char* pinn_backend(void) {
#ifdef PINN_EMU
    return "emu-host";
#else
    return "hip-gfx950";
#endif
}

int pinn_create(const int* layer_dims, int n_layers, int act, int ndims, int nparams, int has_bc, int has_ic,
                const float* dom_lo, const float* dom_hi, float bc_value, pinn_t** out) {
    if (n_layers < 2 || n_layers > PINN_MAX_LAYERS) return fail("n_layers=%d outside [2, %d]", n_layers, PINN_MAX_LAYERS);
    int acts[PINN_MAX_LAYERS];
    for (int a = 0; a < PINN_MAX_LAYERS; ++a) acts[a] = act;
    return pinn_create_ex(layer_dims, n_layers, acts, 0, nullptr, nullptr, ndims, nparams, has_bc, has_ic, dom_lo, dom_hi,
                          bc_value, out);
}

int pinn_create_ex(const int* layer_dims, int n_layers, const int* acts, int n_skips, const int* skip_src,
                   const int* skip_dst, int ndims, int nparams, int has_bc, int has_ic, const float* dom_lo,
                   const float* dom_hi, float bc_value, pinn_t** out) {
    if (!layer_dims || !out || !acts) return fail("null argument");
    if (n_layers < 2 || n_layers > PINN_MAX_LAYERS) return fail("n_layers=%d outside [2, %d]", n_layers, PINN_MAX_LAYERS);
    for (int a = 0; a + 1 < n_layers; ++a)
        if (acts[a] < PINN_ACT_TANH || acts[a] > PINN_ACT_LAST) return fail("unknown activation code %d (activation %d)", acts[a], a);
    if (n_skips < 0 || n_skips > PINN_MAX_SKIPS || (n_skips > 0 && (!skip_src || !skip_dst)))
        return fail("n_skips=%d outside [0, %d]", n_skips, PINN_MAX_SKIPS);
    int dst_of[PINN_MAX_SKIPS] = {0}, src_of[PINN_MAX_SKIPS] = {0}, pre_mask = 0, src_pre_mask = 0;
    for (int k = 0; k < n_skips; ++k) {
        dst_of[k] = skip_dst[k] & ~PINN_SKIP_PRE;
        src_of[k] = skip_src[k] & ~PINN_SKIP_PRE;
        if (skip_dst[k] & PINN_SKIP_PRE) pre_mask |= 1 << k;
        if (skip_src[k] & PINN_SKIP_PRE) src_pre_mask |= 1 << k;
        if (src_of[k] < 0 || src_of[k] >= dst_of[k] || dst_of[k] > n_layers - 2)
            return fail("skip %d: activations %d -> %d outside 0 <= src < dst <= %d", k, src_of[k], dst_of[k], n_layers - 2);
        if (layer_dims[src_of[k] + 1] != layer_dims[dst_of[k] + 1])
            return fail("skip %d joins widths %d and %d", k, layer_dims[src_of[k] + 1], layer_dims[dst_of[k] + 1]);
    }
    // Nesting (round 5): skips may lie inside one another ('fa R fa R fa fa + fa + f': the reference's layout letters pair like brackets)
    // or cross; at most one may START and at most one may END at an activation. A skip during whose life another one opens is "outer":
    // its jets wait in the skip's slab slot, the inner one rides in the registers.
    int outer_mask = 0;
    for (int k = 0; k < n_skips; ++k)
        for (int j = 0; j < n_skips; ++j) {
            if (j == k) continue;
            if (src_of[j] == src_of[k]) return fail("skips %d and %d both start at activation %d", k, j, src_of[k]);
            if (dst_of[j] == dst_of[k]) return fail("skips %d and %d both end at activation %d", k, j, dst_of[k]);
            if (src_of[j] > src_of[k] && src_of[j] < dst_of[k]) outer_mask |= 1 << k;
            // (a skip may start where another one ends only if it does not leave IN FRONT of the activation the other one joins behind)
            if (dst_of[j] == src_of[k] && (src_pre_mask >> k & 1) && !(pre_mask >> j & 1))
                return fail("skip %d starts in front of activation %d, skip %d ends behind it: overlap", k, src_of[k], j);
        }
    const int act = acts[0];
    const int d = ndims + nparams;
    if (ndims < 1 || nparams < 0 || d > PINN_MAX_INPUTS) return fail("ndims+nparams=%d outside [1, %d]", d, PINN_MAX_INPUTS);
    if (layer_dims[0] != d) return fail("layer_dims[0]=%d must equal ndims+nparams=%d", layer_dims[0], d);
    if (layer_dims[n_layers] != 1) return fail("the last layer must have one unit (got %d)", layer_dims[n_layers]);
    int hmax = 0;
    for (int l = 1; l < n_layers; ++l) {
        if (layer_dims[l] < 1) return fail("layer %d has width %d", l, layer_dims[l]);
        if (layer_dims[l] > hmax) hmax = layer_dims[l];
    }
    int hp = round16(hmax);
    if (hp == 48) hp = 64;
    if (hp > 64 && hp < 128) hp = 128;
    if (hp > 128 && hp <= 256) hp = 256;
    if (hp > 256 && hp <= 512) hp = 512;
    if (hp > 512) return fail("hidden width %d > 512 is not supported by this build", hmax);
    const int lh = n_layers - 2;
    // (width 512, round 6: eight bias-gradient rows in the LDS carve -- PinnCfg::ACCB_ROWS -- i.e. up to eight hidden layers)
    if (hp == 512 && lh + 1 > 8) return fail("hidden width %d (padded to 512): at most 8 hidden layers at this width, got %d", hmax, lh + 1);
    pinn_net* net = new (std::nothrow) pinn_net();
    if (!net) return fail("out of memory");
    memset(net, 0, sizeof(*net));
    net->fit_persistent = 2; net->fit_onecu_rounds = 1; net->max_per_cu = 4; net->prepass_in_kernel = 1; net->wgx_chunk_bytes = PINN_WGX_CHUNK_DEFAULT;
    net->n_layers = n_layers; net->act = act; net->ndims = ndims; net->nparams = nparams;
    for (int a = 0; a + 1 < n_layers; ++a) {
        net->act_codes[a >> 4] |= (unsigned long long)acts[a] << (4 * (a & 15));
        if (acts[a] != act) net->act = -1;
        net->act_par[a] = pinn_act_default_param(acts[a]);
    }
    net->n_skips = n_skips;
    for (int k = 0; k < n_skips; ++k) { net->skip_src[k] = src_of[k]; net->skip_dst[k] = dst_of[k]; }
    net->skip_pre = pre_mask; net->skip_src_pre = src_pre_mask; net->skip_outer = outer_mask;
    net->has_bc = has_bc ? 1 : 0; net->has_ic = has_ic ? 1 : 0; net->bc_value = bc_value;
    net->nsp = has_ic ? ndims - 1 : ndims;
    for (int l = 0; l <= n_layers; ++l) net->dims[l] = layer_dims[l];
    for (int i = 0; i < PINN_MAX_INPUTS; ++i) {
        net->lo[i] = (dom_lo && i < ndims) ? dom_lo[i] : 0.0f;
        net->hi[i] = (dom_hi && i < ndims) ? dom_hi[i] : 1.0f;
    }
    pinn_layout_t& L = net->lay;
    L.hp = hp; L.lh = lh; L.d = d;
    L.off_w1 = 0;
    L.off_b1 = hp * d;
    L.off_wh = L.off_b1 + hp;
    L.hidden_stride = hp * hp + hp;
    L.off_wl = L.off_wh + lh * L.hidden_stride;
    L.off_bl = L.off_wl + hp;
    L.off_log_scale = L.off_bl + 1;
    L.off_loss = L.off_bl + 2;
    L.p_core = L.off_bl + 4;
    L.off_extra = L.p_core;
    L.p_total = L.p_core + PINN_EXTRA_SLOTS;
    net->n_cu = 256;
#ifndef PINN_EMU
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        net->n_cu = prop.multiProcessorCount;
#else
    net->n_cu = 2;
#endif
    *out = net;
    return 0;
}

int pinn_destroy(pinn_t* net) {
#ifndef PINN_EMU
    if (net && net->fit_graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)net->fit_graph_exec);
#endif
    delete net;
    return 0;
}

int pinn_layout(const pinn_t* net, pinn_layout_t* out) {
    if (!net || !out) return fail("null argument");
    *out = net->lay;
    return 0;
}

size_t pinn_workspace_bytes(const pinn_t* net, int64_t n_points, int nd, int n2) {
    if (!net) return 0;
    size_t need = 0;
    bool any = false;
    size_t slab_per_wg_max = 0;
    // every form a step of this derivative spec may take: separate second-order streams (affine residual, residual program, the
    // generic path's backward call) and ONE combined second-order stream (affine or program: trace.py lowers to it from a single
    // second derivative on -- `u_t + u_y + u_z = nu u_xx` in four variables exists in that form only), each planned
    // with the general probe and with the arguments of a typical training step (shape-specialised kernels); the largest answer
    const int forms[5][3] = {{PINN_MODE_STEP, PINN_RES_AFFINE, 0}, {PINN_MODE_STEP, PINN_RES_PROGRAM, 0}, {PINN_MODE_BACKWARD, 0, 0},
                             {PINN_MODE_STEP, PINN_RES_AFFINE, 1}, {PINN_MODE_STEP, PINN_RES_PROGRAM, 1}};
    for (int i = 0; i < 10; ++i) {
        Plan plan;
        const int* f = forms[i % 5];
        const bool comb = f[2] != 0;
        if (comb && (pinn_n2(n2) < 1 || nd < 2 || nd > 4 || pinn_n3(n2) > 0)) continue;
        const int n2q = comb ? 1 : n2;
        PinnKArgs typical;
        const bool with_hint = i >= 5 && nd <= net->lay.d;
        if (i >= 5 && !with_hint) continue;
        if (with_hint) typical_step_args(net, &typical, n_points, nd, n2q);
        // (a shape may exist in one form only: nd = 4 with second derivatives runs as the combined stream alone)
        if (make_plan(net, n_points, nd, n2q, &plan, f[0], f[1], comb ? 1 : 0, with_hint ? &typical : nullptr)) continue;
        any = true;
        const size_t v = align256((size_t)plan.rows() * net->lay.p_total * sizeof(float)) +
                         align256(plan.slab_bytes()) + align256(plan.gz_bytes()) +
                         align256((size_t)plan.grid * plan.wt_floats_per_wg * sizeof(float));
        if (v > need) need = v;
        if (plan.slab_vec4_per_wg > slab_per_wg_max) slab_per_wg_max = plan.slab_vec4_per_wg;
    }
    if (!any) return 0;
    // (narrow nets: partial rows of both parities and the private (parameters, exp_avg, exp_avg_sq) of every workgroup of a one-launch fit
    //  chunk, pinn_fit_kernel.h -- 5 x 16 x p_total floats at most)
    //  (+ the saved-jet slabs of the one-CU form's virtual workgroups: at most eight)
    const size_t persist = net->lay.hp <= 32 ? align256(2 * (size_t)PINN_FIT_MAX_WGS * net->lay.p_total * sizeof(float)) +
                                               align256(3 * (size_t)PINN_FIT_MAX_WGS * net->lay.p_total * sizeof(float)) + 512 +
                                               align256(8 * slab_per_wg_max * sizeof(f32x4)) : 0;
    return need + align256((size_t)PINN_MAX_AUX * (size_t)n_points * sizeof(float)) + wt_workspace_bytes(net) +
           wsp_workspace_bytes(net) + 256 + persist;
}

static int jet_forward_impl(pinn_t* net, const float* params, const float* xs, int64_t n_points, const int* dir_cols, int nd,
                            int n2, const float* ic_streams, float ic_const, float* streams_out, void* workspace, size_t workspace_bytes,
                            void* stream) {
    if (!net || !params || !xs || !streams_out) return fail("null argument");
    if (n_points <= 0) return 0;
    if (check_dirs(net, dir_cols, nd, n2)) return 1;
    Plan plan;
    if (make_plan(net, n_points, nd, n2, &plan)) return 1;
    PinnKArgs a;
    fill_args(net, &a, params, xs, n_points, dir_cols, nd, n2, ic_streams, ic_const);
    a.mode = PINN_MODE_FORWARD;
    a.out_streams = streams_out;
    if (net->skip_outer) {
        // nested skips: the outer skip's jets wait in a slab slot between 'R' and '+' -- also in a value-only forward pass
        const size_t need = align256((size_t)plan.grid * plan.slab_vec4_per_wg * sizeof(f32x4));
        if (!workspace || workspace_bytes < need)
            return fail("a net with nested skip connections needs scratch for its forward pass too: call pinn_jet_forward_ws with %zu bytes "
                        "(pinn_workspace_bytes covers it), got %zu", need, workspace ? workspace_bytes : (size_t)0);
        if (((uintptr_t)workspace & 15) != 0) return fail("workspace must be 16-byte aligned");
        a.slab = reinterpret_cast<f32x4*>(workspace);
    }
    set_tile_range(&a, plan, 0, plan.ntiles);
    const int rc = plan.fn(nd, plan.n2k, &a, plan.grid, stream, 0, nullptr);
    note_launch(plan);
    return rc ? fail("tile kernel launch failed (%d)", rc) : 0;
}

int pinn_jet_forward(pinn_t* net, const float* params, const float* xs, int64_t n_points, const int* dir_cols, int nd,
                     int n2, const float* ic_streams, float ic_const, float* streams_out, void* stream) {
    return jet_forward_impl(net, params, xs, n_points, dir_cols, nd, n2, ic_streams, ic_const, streams_out, nullptr, 0, stream);
}

int pinn_jet_forward_ws(pinn_t* net, const float* params, const float* xs, int64_t n_points, const int* dir_cols, int nd,
                        int n2, const float* ic_streams, float ic_const, float* streams_out, void* workspace, size_t workspace_bytes,
                        void* stream) {
    return jet_forward_impl(net, params, xs, n_points, dir_cols, nd, n2, ic_streams, ic_const, streams_out, workspace, workspace_bytes, stream);
}

static int run_train(pinn_t* net, PinnKArgs* a, const Plan& plan, int nd, float* grads, int accumulate, void* workspace,
                     size_t workspace_bytes, void* stream, const pinn_program_t* pre = nullptr,
                     const AdamArgs* adam = nullptr, const double* pre_consts64 = nullptr) {
    const size_t part_bytes = align256((size_t)plan.rows() * net->lay.p_total * sizeof(float));
    const size_t slab_bytes = align256(plan.slab_bytes());
    const size_t aux_bytes = (pre && a->n_aux > 0) ? align256((size_t)a->n_aux * (size_t)a->n_points * sizeof(float)) : 0;
    // transposed hidden weights: one copy written by pinn_transpose_kernel (widths >= 128) or one scratch per workgroup
    // filled by the tile kernel itself (slab-in-LDS kernels)
    const size_t wt_bytes = plan.wt_floats_per_wg ? align256((size_t)plan.grid * plan.wt_floats_per_wg * sizeof(float))
                                                  : (plan.wt_global ? wt_workspace_bytes(net) : 0);
    const size_t gz_bytes = align256(plan.gz_bytes());
    const size_t wsp_bytes = plan.split ? wsp_workspace_bytes(net) : 0;
    const size_t need = part_bytes + slab_bytes + aux_bytes + wt_bytes + gz_bytes + wsp_bytes;
    if (!workspace || workspace_bytes < need)
        return fail("workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    if (((uintptr_t)workspace & 15) != 0) return fail("workspace must be 16-byte aligned");
    char* ws = reinterpret_cast<char*>(workspace);
    a->prof = g_phase_prof;
    a->debug_flags = g_pinn_debug_flags;
    a->partials = reinterpret_cast<float*>(ws);
    a->slab = reinterpret_cast<f32x4*>(ws + part_bytes);
    a->gzslab = gz_bytes ? reinterpret_cast<f32x4*>(ws + part_bytes + slab_bytes + aux_bytes + wt_bytes) : nullptr;
    a->partial_row0 = plan.grid;
    if (aux_bytes) {
        float* aux = reinterpret_cast<float*>(ws + part_bytes + slab_bytes);
        a->aux = aux;
        // the pre-pass runs in fp64 (include/pinn.h, pinn_residual_t::pre_consts64): its constants unrounded
        PinnPreConsts pc64;
        for (int k = 0; k < PINN_MAX_CONSTS; ++k) pc64.v[k] = pre_consts64 ? pre_consts64[k] : (double)pre->consts[k];
        int nregs = a->d;
        for (int i = 0; i < pre->n_ops; ++i) {
            const uint32_t w = pre->code[i];
            const int op = w & 255, dst = (w >> 8) & 255;
            if (op != PINN_OP_STORE && dst + 1 > nregs) nregs = dst + 1;
        }
        // in the kernel its registers are doubles in the (not yet used) LDS activation buffers: one tile's worth of points must fit --
        // always, except for a long program in front of the narrowest one-stream kernels, which takes the separate launch
        if (net->prepass_in_kernel && 2 * nregs <= plan.prepass_floats_per_lane) {
            a->pre = *pre;          // evaluated in the prologue of the tile kernel: one launch (and one dependent-launch gap) less
            a->pre_consts64 = pc64;
            a->pre_nregs = nregs;
        } else {
            const int blocks = (int)((a->n_points + 255) / 256);
#ifdef PINN_EMU
            emu::launch(blocks, 256, 0, [&] { pinn_aux_kernel(a->xs, a->n_points, a->d, *pre, pc64, aux); });
#else
            hipLaunchKernelGGL(pinn_aux_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a->xs, a->n_points, a->d,
                               *pre, pc64, aux);
            if (hipGetLastError() != hipSuccess) return fail("pre-pass kernel launch failed");
#endif
        }
    }
    if (plan.wt_floats_per_wg)
        a->wt = reinterpret_cast<float*>(ws + part_bytes + slab_bytes + aux_bytes);
    else if (wt_bytes) {
        // widths >= 128: transposed copy of the hidden weights for the data-gradient GEMM (the weights change every step)
        float* wt = reinterpret_cast<float*>(ws + part_bytes + slab_bytes + aux_bytes);
        a->wt = wt;
        const int hp = net->lay.hp, blocks = (hp / 32) * (hp / 32) * net->lay.lh;
        const float* wh = a->params + net->lay.off_wh;
        const int stride = net->lay.hidden_stride;
#ifdef PINN_EMU
        emu::launch(blocks, 256, 32 * 33 * sizeof(float), [&] { pinn_transpose_kernel(wh, stride, hp, wt); });
#else
        hipLaunchKernelGGL(pinn_transpose_kernel, dim3(blocks), dim3(256), 32 * 33 * sizeof(float), (hipStream_t)stream, wh,
                           stride, hp, wt);
        if (hipGetLastError() != hipSuccess) return fail("transpose kernel launch failed");
#endif
    }
    if (wsp_bytes && net->lay.lh > 0) {
        // split-bf16 kernels: hidden weights -> hi / mid / lo bf16 MFMA fragments (the weights change every step)
        pinn_s16x8* wsp = reinterpret_cast<pinn_s16x8*>(ws + part_bytes + slab_bytes + aux_bytes + wt_bytes + gz_bytes);
        a->wsp = wsp;
        const int hp = net->lay.hp, lh = net->lay.lh;
        const int blocks = (lh * 2 * (hp / 32) * (hp / 16) * 64 + 255) / 256;
        const float* wh = a->params + net->lay.off_wh;
        const int stride = net->lay.hidden_stride;
#ifdef PINN_EMU
        emu::launch(blocks, 256, 0, [&] { pinn_wsplit_kernel(wh, stride, hp, lh, wsp); });
#else
        hipLaunchKernelGGL(pinn_wsplit_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, wh, stride, hp, lh, wsp);
        if (hipGetLastError() != hipSuccess) return fail("weight-split kernel launch failed");
#endif
    }
#ifndef PINN_EMU
    ProfEvents* pe = g_profile ? prof_events() : nullptr;
    if (g_profile && !pe) return fail("hipEventCreate failed");
    if (pe) pe->have_wgrad = false;
#endif
    if (g_fit_persist.active) {
        // a whole chunk of fit iterations in ONE launch (pinn_fit_kernel.h): the step must be one pass of a non-streamed kernel, the
        // instantiation must carry the kernel, the scratch for rows / private states must fit behind `need`. Mode 2: the one-CU form
        // (virtual workgroups, no device-scope wait) for batches of at most fit_onecu_rounds sweeps of its virtual workgroups; mode 1:
        // the grid form (every workgroup resident).
        g_fit_persist.done = 0;
        long long has[3] = {0, 0, 0};
        const size_t pc = (size_t)net->lay.p_total;
        bool ok = !plan.wgx && !plan.wt_global && !plan.split && plan.chunk_tiles >= plan.ntiles && !g_profile && adam &&
                  (!aux_bytes || net->prepass_in_kernel) && plan.fn(nd, plan.n2k, a, plan.grid, stream, 3, has) == 0 && has[0];
        const int vw = (int)has[1];
        const bool onecu = ok && net->fit_persistent == 2 && vw >= 2 && vw <= 8 && plan.ntiles <= (int64_t)vw * net->fit_onecu_rounds;
        const bool gridform = ok && net->fit_persistent == 1 && plan.grid <= PINN_FIT_MAX_WGS;
        const int rows = onecu ? vw : plan.grid, states = onecu ? 1 : plan.grid;
        const size_t rows_b = align256(2 * (size_t)rows * pc * sizeof(float)), state_b = align256(3 * (size_t)states * pc * sizeof(float));
        const size_t slab1_b = onecu ? align256((size_t)vw * plan.slab_vec4_per_wg * sizeof(f32x4)) : 0;      // the virtual workgroups' own slabs
        if ((onecu || gridform) && workspace_bytes >= need + rows_b + state_b + 256 + slab1_b) {
            PinnFitP& P = g_fit_persist.p;
            P.rows = reinterpret_cast<float*>(ws + need);
            P.state = reinterpret_cast<float*>(ws + need + rows_b);
            P.sync = reinterpret_cast<unsigned*>(ws + need + rows_b + state_b);
            if (onecu && a->slab) a->slab = reinterpret_cast<f32x4*>(ws + need + rows_b + state_b + 256);
            P.params = adam->params; P.m = adam->m; P.v = adam->v; P.mask = adam->mask; P.step_ptr = adam->step_ptr; P.grads = grads;
            P.b1 = adam->b1; P.b2 = adam->b2; P.eps = adam->eps; P.off_loss = net->lay.off_loss;
#ifdef PINN_EMU
            memset(P.sync, 0, 256);
#else
            if (!onecu && hipMemsetAsync(P.sync, 0, 256, (hipStream_t)stream) != hipSuccess) return fail("hipMemsetAsync failed");
#endif
            set_tile_range(a, plan, 0, plan.ntiles);
            const int rc = plan.fn(nd, plan.n2k, a, onecu ? 1 : plan.grid, stream, onecu ? 4 : 2, reinterpret_cast<long long*>(&P));
            note_launch(plan);
            if (rc == 1) return 0;               // (this instantiation has no such form after all: the caller falls back)
            if (rc) return fail("fit kernel launch failed (%d)", rc);
            g_fit_persist.done = 1;
            g_fit_last_sync = onecu ? nullptr : P.sync;        // (grid form: its timeout flag, read back by pinn_fit_chunk_status)
        }
        return 0;
    }
    // one pass (any non-WGX kernel; a WGX batch whose slabs fit the net's wgx_chunk_bytes) or chunk by chunk: tile kernel ->
    // weight-gradient kernel -> reduction of the partial rows, later chunks ADD into `grads`, Adam rides in the last reduction
    for (int64_t t0 = 0; t0 < plan.ntiles || t0 == 0; t0 += plan.chunk_tiles) {
        const bool last = t0 + plan.chunk_tiles >= plan.ntiles;
        set_tile_range(a, plan, t0, t0 + plan.chunk_tiles);
#ifndef PINN_EMU
        if (pe) hipEventRecord(pe->tile0, (hipStream_t)stream);
#endif
        const int rc = plan.fn(nd, plan.n2k, a, plan.grid, stream, 0, nullptr);
        note_launch(plan);
        if (rc) return fail("tile kernel launch failed (%d)", rc);
#ifndef PINN_EMU
        if (pe) { hipEventRecord(pe->tile1, (hipStream_t)stream); pe->have_tile = true; }
#endif
        if (plan.wgx && net->lay.lh > 0) {
#ifndef PINN_EMU
            if (pe) hipEventRecord(pe->wg0, (hipStream_t)stream);
#endif
            const int rw = plan.wfn(nd, plan.n2k, plan.comb, plan.mt, a, plan.grid2, stream, 0, nullptr);
            if (rw) return fail("weight-gradient kernel launch failed (%d)", rw);
#ifndef PINN_EMU
            if (pe) { hipEventRecord(pe->wg1, (hipStream_t)stream); pe->have_wgrad = true; }
#endif
        }
        const int rows = (plan.wgx && net->lay.lh > 0) ? plan.rows() : plan.grid;
        if (launch_reduce(a->partials, rows, net->lay.p_total, grads, (accumulate || t0 > 0) ? 1 : 0, stream, last ? adam : nullptr))
            return 1;
        if (last) break;
    }
    return 0;
}

int pinn_jet_backward(pinn_t* net, const float* params, const float* xs, int64_t n_points, const int* dir_cols, int nd,
                      int n2, const float* ic_streams, float ic_const, const float* grad_streams, float* grads,
                      int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!net || !params || !xs || !grad_streams || !grads) return fail("null argument");
    if (n_points <= 0) return 0;
    if (check_dirs(net, dir_cols, nd, n2)) return 1;
    Plan plan;
    if (make_plan(net, n_points, nd, n2, &plan, PINN_MODE_BACKWARD, 0)) return 1;
    PinnKArgs a;
    fill_args(net, &a, params, xs, n_points, dir_cols, nd, n2, ic_streams, ic_const);
    a.mode = PINN_MODE_BACKWARD;
    a.gin = grad_streams;
    return run_train(net, &a, plan, nd, grads, accumulate, workspace, workspace_bytes, stream);
}

static int check_program(const pinn_program_t& pg, int first_temp, int n_consts_max, bool pre, int n_aux, const char* what) {
    if (pg.n_ops < 0 || pg.n_ops > PINN_MAX_OPS || pg.n_consts < 0 || pg.n_consts > n_consts_max)
        return fail("%s program size out of range (ops=%d consts=%d)", what, pg.n_ops, pg.n_consts);
    for (int i = 0; i < pg.n_ops; ++i) {
        const uint32_t w = pg.code[i];
        const int op = w & 255, dst = (w >> 8) & 255, ra = (w >> 16) & 255, rb = (w >> 24) & 255;
        if (op > PINN_OP_STORE) return fail("%s program instruction %d: unknown opcode %d", what, i, op);
        if (op == PINN_OP_STORE) {
            if (!pre) return fail("%s program instruction %d: STORE is a pre-pass instruction", what, i);
            if (rb >= n_aux || ra >= PINN_MAX_REGS) return fail("%s program instruction %d: bad STORE", what, i);
            continue;
        }
        const bool b_is_reg = op == PINN_OP_ADD || op == PINN_OP_SUB || op == PINN_OP_MUL || op == PINN_OP_DIV;
        if (dst < first_temp) return fail("%s program instruction %d overwrites an input register", what, i);
        if (dst >= PINN_MAX_REGS || (op != PINN_OP_CONST && ra >= PINN_MAX_REGS) || (b_is_reg && rb >= PINN_MAX_REGS))
            return fail("%s program instruction %d uses a register >= %d", what, i, PINN_MAX_REGS);
        if (op == PINN_OP_CONST && ra >= pg.n_consts) return fail("%s program instruction %d: bad constant", what, i);
        if (op == PINN_OP_POW && rb >= pg.n_consts) return fail("%s program instruction %d: bad exponent", what, i);
    }
    return 0;
}

static int residual_step_impl(pinn_t* net, const pinn_residual_t* residual, const float* params, const float* xs,
                              int64_t n_points, const int* dir_cols, int nd, int n2, const float* ic_streams,
                              float ic_const, float inv_n_global, float* grads, void* workspace, size_t workspace_bytes,
                              void* stream, const AdamArgs* adam, int accumulate = 0) {
    if (!net || !residual || !params || !xs || !grads) return fail("null argument");
    if (n_points <= 0) return fail("n_points must be positive");
    if (check_dirs(net, dir_cols, nd, n2)) return 1;
    if (residual->n_aux < 0 || residual->n_aux > PINN_MAX_AUX) return fail("n_aux=%d outside [0, %d]", residual->n_aux, PINN_MAX_AUX);
    if (residual->kind != PINN_RES_AFFINE && residual->kind != PINN_RES_PROGRAM) return fail("unknown residual kind %d", residual->kind);
    Plan plan;
    const int comb = residual->combined ? 1 : 0;
    // (the plan itself is made below, with the call's own arguments; here only the stream count of the instantiation)
    const int n2k = comb ? 1 : pick_n2(nd, n2);
    if (comb && (n2 != 1 || nd < 2 || nd > 4)) return fail("combined second-order stream needs n2 == 1 and nd in {2, 3, 4}");
    if (n2k < 0) return fail("unsupported derivative spec nd=%d n2=%d n3=%d", nd, pinn_n2(n2), pinn_n3(n2));
    const int d = net->lay.d;
    const int s_user = pinn_ns(nd, n2), s_kernel = pinn_ns(nd, n2k), shift = s_kernel - s_user;
    PinnKArgs a;
    fill_args(net, &a, params, xs, n_points, dir_cols, nd, n2, ic_streams, ic_const);
    a.res_kind = residual->kind;
    a.n_aux = residual->n_aux;
    a.n_vars = residual->n_vars;
    if (a.n_vars < 0 || a.n_vars > PINN_MAX_VARS || a.n_vars > PINN_EXTRA_SLOTS)
        return fail("n_vars=%d outside [0, %d]", a.n_vars, PINN_MAX_VARS);
    if (a.n_vars > 0 && residual->kind != PINN_RES_PROGRAM) return fail("trainable variables need a residual program (kind PINN_RES_PROGRAM)");
    a.ic_var1 = residual->ic_var1;
    if (a.ic_var1 < 0 || a.ic_var1 > PINN_EXTRA_SLOTS) return fail("ic_var1=%d outside [0, %d]", a.ic_var1, PINN_EXTRA_SLOTS);
    if (a.ic_var1 > 0 && (!net->has_ic || ic_streams)) return fail("ic_var1 needs a problem with an initial condition and no ic_streams");
    a.comb = comb;
    for (int k = 0; k < PINN_MAX_DIRS; ++k) a.comb_w[k] = (comb && k < nd) ? residual->comb_w[k] : 0.0f;
    a.ic_rows = residual->ic_rows ? 1 : 0;
    if (a.ic_rows) {
        if (!net->has_ic || ic_streams || residual->ic_var1 > 0) return fail("ic_rows needs a problem with an initial condition, no ic_streams and no ic_var1");
        for (int s = 0; s < PINN_MAX_STREAMS; ++s) {
            a.ic_row[s] = (s < s_user) ? residual->ic_row[s] : -1;
            a.ic_cst[s] = (s < s_user) ? residual->ic_cst[s] : 0.0f;
            if (a.ic_row[s] >= residual->n_aux) return fail("ic_row[%d] refers to a missing pre-pass row", s);
        }
    }
    if (residual->n_aux > 0 && check_program(residual->pre, d, PINN_MAX_CONSTS, true, residual->n_aux, "pre-pass")) return 1;
    if (residual->kind == PINN_RES_AFFINE) {
        for (int s = 0; s < PINN_MAX_STREAMS; ++s) {
            a.coef[s] = (s < s_user) ? residual->coef[s] : 0.0f;
            a.coef_row[s] = (s < s_user) ? residual->coef_row[s] : -1;
            if (a.coef_row[s] >= residual->n_aux) return fail("coef_row[%d] refers to a missing pre-pass row", s);
        }
        a.src_const = residual->src_const;
        a.src_row = residual->src_row;
        if (a.src_row >= residual->n_aux) return fail("src_row refers to a missing pre-pass row");
    } else {
        const pinn_program_t* program = &residual->program;
        if (program->n_ops < 1) return fail("empty residual program");
        if (check_program(*program, s_user + d + residual->n_aux + residual->n_vars, PINN_MAX_CONSTS, false, 0, "residual")) return 1;
        // the program addresses registers with the caller's stream count; re-base everything behind the streams
        // onto the instantiation's stream count (extra second-derivative streams sit in between)
        a.prog = *program;
        for (int i = 0; i < program->n_ops; ++i) {
            const uint32_t w = program->code[i];
            int op = w & 255, dst = (w >> 8) & 255, ra = (w >> 16) & 255, rb = (w >> 24) & 255;
            auto fix = [&](int r) { return r >= s_user ? r + shift : r; };
            const bool b_is_reg = op == PINN_OP_ADD || op == PINN_OP_SUB || op == PINN_OP_MUL || op == PINN_OP_DIV;
            dst = fix(dst);
            if (op != PINN_OP_CONST) ra = fix(ra);
            if (b_is_reg) rb = fix(rb);
            if (dst >= PINN_MAX_REGS || ra >= PINN_MAX_REGS || (b_is_reg && rb >= PINN_MAX_REGS))
                return fail("program instruction %d uses a register >= %d after re-basing", i, PINN_MAX_REGS);
            a.prog.code[i] = (uint32_t)op | ((uint32_t)dst << 8) | ((uint32_t)ra << 16) | ((uint32_t)rb << 24);
        }
    }
    a.mode = PINN_MODE_STEP;
    a.inv_n = inv_n_global;
    if (make_plan(net, n_points, nd, n2, &plan, PINN_MODE_STEP, residual->kind, comb, &a)) return 1;      // with the call's own arguments
    return run_train(net, &a, plan, nd, grads, accumulate, workspace, workspace_bytes, stream, &residual->pre, adam, residual->pre_consts64);
}

int pinn_residual_step(pinn_t* net, const pinn_residual_t* residual, const float* params, const float* xs,
                       int64_t n_points, const int* dir_cols, int nd, int n2, const float* ic_streams, float ic_const,
                       float inv_n_global, float* grads, void* workspace, size_t workspace_bytes, void* stream) {
    return residual_step_impl(net, residual, params, xs, n_points, dir_cols, nd, n2, ic_streams, ic_const, inv_n_global,
                              grads, workspace, workspace_bytes, stream, nullptr);
}

int pinn_residual_step_add(pinn_t* net, const pinn_residual_t* residual, const float* params, const float* xs,
                           int64_t n_points, const int* dir_cols, int nd, int n2, const float* ic_streams, float ic_const,
                           float inv_n_global, float* grads, void* workspace, size_t workspace_bytes, void* stream) {
    return residual_step_impl(net, residual, params, xs, n_points, dir_cols, nd, n2, ic_streams, ic_const, inv_n_global,
                              grads, workspace, workspace_bytes, stream, nullptr, 1);
}

int pinn_residual_adam_step(pinn_t* net, const pinn_residual_t* residual, float* params, const float* xs,
                            int64_t n_points, const int* dir_cols, int nd, int n2, const float* ic_streams,
                            float ic_const, float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* mask,
                            int32_t* step_ptr, int32_t step, float lr, float beta1, float beta2, float eps,
                            float* loss_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!net || !exp_avg || !exp_avg_sq || !step_ptr) return fail("null argument");
    if (step < 1) return fail("step must be >= 1");
    AdamArgs adam = {params, exp_avg, exp_avg_sq, mask, step_ptr, step, lr, beta1, beta2, eps, loss_out, net->lay.off_loss};
    return residual_step_impl(net, residual, params, xs, n_points, dir_cols, nd, n2, ic_streams, ic_const,
                              1.0f / (float)n_points, grads, workspace, workspace_bytes, stream, &adam);
}

// the sampler description of a fit chunk in the form the reduction's tail takes (non-zero: columns it cannot draw)
static int fit_next_spec(const pinn_net* net, float* xs, int64_t n_points, const int* kind, const float* a, const float* b, uint64_t seed,
                         PinnNextBatch* out) {
    if (!net || !xs || !kind || !a || !b || n_points <= 0 || net->lay.d < 1 || net->lay.d > PINN_MAX_INPUTS) return 1;
    memset(out, 0, sizeof(*out));
    out->xs = xs; out->n = (long long)n_points; out->spec.d = net->lay.d;
    for (int c = 0; c < net->lay.d; ++c) {
        if (kind[c] < PINN_SAMPLE_UNIFORM || kind[c] > PINN_SAMPLE_CONST) return 1;
        out->spec.kind[c] = kind[c]; out->spec.a[c] = a[c]; out->spec.b[c] = b[c];
    }
    out->k0 = (unsigned)(seed & 0xffffffffull); out->k1 = (unsigned)(seed >> 32);
    return 0;
}

// does a fused step of this batch run as ONE tile pass + ONE reduction (not chunk by chunk)?
static bool single_pass_step(pinn_net* net, const pinn_residual_t* residual, int64_t n_points, int nd, int n2) {
    if (!residual) return false;
    Plan plan;
    const int comb = residual->combined ? 1 : 0;
    if (make_plan(net, n_points, nd, n2, &plan, PINN_MODE_STEP, residual->kind, comb, nullptr)) return false;
    return plan.chunk_tiles >= plan.ntiles;
}

int pinn_fit_steps(pinn_t* net, const pinn_residual_t* residual, float* params, float* xs, int64_t n_points,
                   const int* kind, const float* a, const float* b, uint64_t seed, uint64_t call_index0,
                   const int* dir_cols, int nd, int n2, float ic_const, float* grads, float* exp_avg, float* exp_avg_sq,
                   const uint8_t* mask, int32_t* step_ptr, int32_t step0, float lr, float beta1, float beta2, float eps,
                   float* loss_history, int32_t k_steps, void* workspace, size_t workspace_bytes, void* stream) {
    if (!net || !xs || !loss_history) return fail("null argument");
    if (k_steps < 0 || step0 < 1) return fail("k_steps must be >= 0 and step0 >= 1");
    // K iterations of the reference's fit loop (model_torch.py:426-464) enqueued by ONE call: sample, fused step, Adam. Nothing
    // here waits for the device; the arguments that change from one iteration to the next (Philox batch counter, Adam step,
    // slot of the loss history) travel by value, so a launch graph would have to be re-instantiated per iteration anyway
    // from the second iteration on the batch is drawn by the reduction launch of the iteration before (pinn_reduce_kernel's tail: the
    // tile kernel of that iteration is through with the buffer), which saves every iteration one dependent launch -- where the step
    // is one pass (a chunked WGX step reduces several times per iteration: those keep the sampler launch)
    PinnNextBatch next = {nullptr, 0, {}, 0u, 0u, 0ull};
    const bool hand_over = fit_next_spec(net, xs, n_points, kind, a, b, seed, &next) == 0 && single_pass_step(net, residual, n_points, nd, n2);
    int rc = 0;
    for (int32_t k = 0; k < k_steps && !rc; ++k) {
        if (k == 0 || !hand_over) rc = pinn_sample_points(xs, n_points, net->lay.d, kind, a, b, seed, call_index0 + (uint64_t)k, stream);
        if (rc) break;
        if (hand_over && k + 1 < k_steps) { next.call = call_index0 + (uint64_t)k + 1; g_fit_next = next; }
        rc = pinn_residual_adam_step(net, residual, params, xs, n_points, dir_cols, nd, n2, nullptr, ic_const, grads, exp_avg,
                                     exp_avg_sq, mask, step_ptr, step0 + k, lr, beta1, beta2, eps, loss_history + k, workspace,
                                     workspace_bytes, stream);
        g_fit_next.n = 0;
    }
    return rc;
}

size_t pinn_fit_ctrl_bytes(void) { return sizeof(PinnFitCtrl); }

static int g_fit_graph_stats[4] = {0, 0, 0, 0};      // chunks replayed, graphs captured, captures refused, last HIP error of a refusal
int pinn_debug_fit_graph_stats(int32_t out[4]) {
    if (!out) return fail("null argument");
    for (int i = 0; i < 4; ++i) out[i] = g_fit_graph_stats[i];
    return 0;
}

// FNV-1a over the bytes of everything a captured chunk depends on
static void key_mix(unsigned long long (&key)[4], const void* p, size_t n) {
    const unsigned char* b = reinterpret_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) {
        unsigned long long& h = key[i & 3];
        h = (h ^ b[i]) * 1099511628211ull;
    }
}

int pinn_fit_steps_graph(pinn_t* net, const pinn_residual_t* residual, float* params, float* xs, int64_t n_points,
                         const int* kind, const float* a, const float* b, uint64_t seed, uint64_t call_index0,
                         const int* dir_cols, int nd, int n2, float ic_const, float* grads, float* exp_avg, float* exp_avg_sq,
                         const uint8_t* mask, int32_t* step_ptr, int32_t step0, float lr, float beta1, float beta2, float eps,
                         float* loss_history, int32_t k_steps, void* workspace, size_t workspace_bytes, void* ctrl,
                         size_t ctrl_bytes, void* stream) {
    if (!net || !residual || !xs || !loss_history || !kind || !a || !b || !dir_cols) return fail("null argument");
    if (k_steps < 0 || step0 < 1) return fail("k_steps must be >= 0 and step0 >= 1");
    // narrow nets, batches of a few tiles: the whole chunk -- any length up to PINN_FIT_CHUNK_MAX -- as ONE launch (pinn_fit_kernel.h);
    // the eager loop's trajectory to fp32 round-off
    if (net->fit_persistent && net->lay.hp <= 32 && ctrl && ctrl_bytes >= sizeof(PinnFitCtrl) && k_steps >= 1 && k_steps <= PINN_FIT_CHUNK_MAX &&
        !g_profile && !g_phase_prof && residual && params && grads && exp_avg && exp_avg_sq && step_ptr) {
        PinnNextBatch spec_probe = {nullptr, 0, {}, 0u, 0u, 0ull};
        if (fit_next_spec(net, xs, n_points, kind, a, b, seed, &spec_probe) == 0) {
            PinnFitCtrl* dctrl = reinterpret_cast<PinnFitCtrl*>(ctrl);
            PinnFitCtrlArgs ca;
            ca.c.call_index0 = call_index0; ca.c.loss_base = loss_history; ca.c.step0 = step0; ca.c.pad = 0;
            ca.c.k0 = (unsigned)(seed & 0xffffffffull); ca.c.k1 = (unsigned)(seed >> 32);
            for (int k = 0; k < PINN_FIT_CHUNK_MAX; ++k) {
                ca.c.step_size[k] = 0.0f; ca.c.bc2_sqrt[k] = 1.0f;
                if (k < k_steps) pinn_adam_scalars((double)(step0 + k), lr, beta1, beta2, &ca.c.step_size[k], &ca.c.bc2_sqrt[k]);
            }
            memset(&g_fit_persist, 0, sizeof(g_fit_persist));
            PinnFitP& P = g_fit_persist.p;
            P.ctrl = dctrl; P.k_steps = k_steps; P.xs = xs; P.n = (long long)n_points; P.spec = spec_probe.spec;
            g_fit_persist.active = 1;
#ifdef PINN_EMU
            emu::launch(1, 128, 0, [&] { pinn_fit_ctrl_kernel(dctrl, ca); });
            const int rc0 = 0;
#else
            hipLaunchKernelGGL(pinn_fit_ctrl_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, dctrl, ca);
            const int rc0 = (hipGetLastError() != hipSuccess) ? fail("control-block launch failed") : 0;
#endif
            const int rc = rc0 ? rc0
                           : pinn_residual_adam_step(net, residual, params, xs, n_points, dir_cols, nd, n2, nullptr, ic_const, grads, exp_avg,
                                                     exp_avg_sq, mask, step_ptr, step0, lr, beta1, beta2, eps, loss_history, workspace,
                                                     workspace_bytes, stream);
            const int did = g_fit_persist.done;
            g_fit_persist.active = 0;
            if (rc) return rc;
            if (did) { ++g_fit_graph_stats[0]; return 0; }          // (counted with the replayed chunks: one launch for the chunk)
        }
    }
#ifdef PINN_EMU
    return pinn_fit_steps(net, residual, params, xs, n_points, kind, a, b, seed, call_index0, dir_cols, nd, n2, ic_const, grads, exp_avg,
                          exp_avg_sq, mask, step_ptr, step0, lr, beta1, beta2, eps, loss_history, k_steps, workspace, workspace_bytes, stream);
#else
    // the graph pays where launch gaps are a visible share of an iteration; chunks that do not qualify run the eager loop
    // (only whole chunks: the tail of a fit would capture a graph of its own that nothing replays)
    if (k_steps != PINN_FIT_CHUNK_MAX || !ctrl || ctrl_bytes < sizeof(PinnFitCtrl) || g_profile || g_phase_prof)
        return pinn_fit_steps(net, residual, params, xs, n_points, kind, a, b, seed, call_index0, dir_cols, nd, n2, ic_const, grads, exp_avg,
                              exp_avg_sq, mask, step_ptr, step0, lr, beta1, beta2, eps, loss_history, k_steps, workspace, workspace_bytes, stream);
    // everything the captured launches carry BY VALUE or by address (the per-iteration values travel through the control block)
    unsigned long long key[4] = {14695981039346656037ull, 14695981039346656037ull ^ 1, 14695981039346656037ull ^ 2, 14695981039346656037ull ^ 3};
    key_mix(key, residual, sizeof(*residual));
    const void* ptrs[] = {params, xs, grads, exp_avg, exp_avg_sq, mask, step_ptr, workspace, ctrl, stream};
    key_mix(key, ptrs, sizeof(ptrs));
    // (not the sampler's key `seed`: it travels through the control block, so the chunk recorded by one fit call is replayed by the next)
    const long long ints[] = {(long long)n_points, nd, n2, (long long)workspace_bytes, net->gemm_mode, net->tanh_mode, net->max_per_cu,
                              net->prepass_in_kernel, (long long)net->wgx_chunk_bytes};
    key_mix(key, ints, sizeof(ints));
    const float flts[] = {ic_const, lr, beta1, beta2, eps};
    key_mix(key, flts, sizeof(flts));
    key_mix(key, kind, sizeof(int) * net->lay.d); key_mix(key, a, sizeof(float) * net->lay.d); key_mix(key, b, sizeof(float) * net->lay.d);
    key_mix(key, dir_cols, sizeof(int) * (nd > 0 ? nd : 0));
    int dev = 0;
    (void)hipGetDevice(&dev);
    key_mix(key, &dev, sizeof(dev));
    PinnFitCtrl* dctrl = reinterpret_cast<PinnFitCtrl*>(ctrl);
    hipStream_t hs = (hipStream_t)stream;
    const bool hit = net->fit_graph_exec && net->fit_graph_k == k_steps && memcmp(key, net->fit_graph_key, sizeof(key)) == 0;
    if (!hit) {
        if (net->fit_graph_exec) { (void)hipGraphExecDestroy((hipGraphExec_t)net->fit_graph_exec); net->fit_graph_exec = nullptr; }
        // one eager iteration's worth of planning has to have happened (function attributes, occupancy queries are not capturable):
        // the first chunk of a configuration runs eagerly AND is followed by the capture of the graph the next chunks replay
        const int rc = pinn_fit_steps(net, residual, params, xs, n_points, kind, a, b, seed, call_index0, dir_cols, nd, n2, ic_const, grads,
                                      exp_avg, exp_avg_sq, mask, step_ptr, step0, lr, beta1, beta2, eps, loss_history, k_steps, workspace,
                                      workspace_bytes, stream);
        if (rc) return rc;
        hipGraph_t graph = nullptr;
        // capture happens on a stream of the library's own (torch's current stream is usually the legacy default stream, which cannot
        // capture); nothing EXECUTES on it -- the nodes are kernel launches without a stream of their own, the graph is launched on `hs`
        static thread_local hipStream_t cap_streams[64] = {nullptr};
        hipStream_t& cs = cap_streams[dev & 63];
        if (!cs && hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ++g_fit_graph_stats[2]; return 0; }
        const hipError_t be = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
        if (be != hipSuccess) { (void)hipGetLastError(); ++g_fit_graph_stats[2]; g_fit_graph_stats[3] = (int)be; return 0; }
        int crc = 0;
        PinnNextBatch cap_next = {nullptr, 0, {}, 0u, 0u, 0ull};
        const bool cap_hand_over = fit_next_spec(net, xs, n_points, kind, a, b, seed, &cap_next) == 0 && single_pass_step(net, residual, n_points, nd, n2);
        for (int32_t k = 0; k < k_steps && !crc; ++k) {
            g_fit_capture = {dctrl, k};
            // (the by-value step numbers and loss slots of the captured launches are placeholders: the kernels read the control block)
            if (k == 0 || !cap_hand_over) crc = pinn_sample_points(xs, n_points, net->lay.d, kind, a, b, seed, 0, cs);
            if (cap_hand_over && k + 1 < k_steps) g_fit_next = cap_next;
            if (!crc) crc = pinn_residual_adam_step(net, residual, params, xs, n_points, dir_cols, nd, n2, nullptr, ic_const, grads, exp_avg,
                                                    exp_avg_sq, mask, step_ptr, 1, lr, beta1, beta2, eps, loss_history, workspace,
                                                    workspace_bytes, cs);
            g_fit_next.n = 0;       // (as in the eager loop: the LAST reduction of a chunk draws nothing -- ADVICE r4)
        }
        g_fit_capture = {nullptr, 0};
        g_fit_next.n = 0;
        const hipError_t ce = hipStreamEndCapture(cs, &graph);
        if (crc || ce != hipSuccess || !graph) {
            (void)hipGetLastError(); ++g_fit_graph_stats[2]; g_fit_graph_stats[3] = crc ? -crc : (int)ce;
            if (graph) (void)hipGraphDestroy(graph);
            return crc ? crc : 0;
        }
        hipGraphExec_t exec = nullptr;
        const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) { (void)hipGetLastError(); ++g_fit_graph_stats[2]; g_fit_graph_stats[3] = (int)ie; (void)hipGraphDestroy(graph); return 0; }
        (void)hipGraphDestroy(graph);
        net->fit_graph_exec = exec;
        ++g_fit_graph_stats[1];
        net->fit_graph_k = k_steps;
        memcpy(net->fit_graph_key, key, sizeof(key));
        return 0;                   // (this chunk ran eagerly above)
    }
    PinnFitCtrlArgs ca;
    ca.c.call_index0 = call_index0; ca.c.loss_base = loss_history; ca.c.step0 = step0; ca.c.pad = 0;
    ca.c.k0 = (unsigned)(seed & 0xffffffffull); ca.c.k1 = (unsigned)(seed >> 32);
    for (int k = 0; k < PINN_FIT_CHUNK_MAX; ++k) {
        ca.c.step_size[k] = 0.0f; ca.c.bc2_sqrt[k] = 1.0f;
        if (k < k_steps) pinn_adam_scalars((double)(step0 + k), lr, beta1, beta2, &ca.c.step_size[k], &ca.c.bc2_sqrt[k]);
    }
    hipLaunchKernelGGL(pinn_fit_ctrl_kernel, dim3(1), dim3(128), 0, hs, dctrl, ca);
    if (hipGetLastError() != hipSuccess) return fail("control-block launch failed");
    if (hipGraphLaunch((hipGraphExec_t)net->fit_graph_exec, hs) != hipSuccess) return fail("hipGraphLaunch failed");
    ++g_fit_graph_stats[0];
    return 0;
#endif
}

static int adam_launch(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* mask, int64_t n,
                       int32_t* step_ptr, int32_t step, float lr, float beta1, float beta2, float eps, void* stream,
                       float* loss_out = nullptr, int off_loss = -1) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !step_ptr) return fail("null argument");
    if (n <= 0) return 0;
    const int blocks = (int)((n + 255) / 256);
    float step_size = 0.0f, bc2_sqrt = 1.0f;
    if (step > 0) pinn_adam_scalars((double)step, lr, beta1, beta2, &step_size, &bc2_sqrt);
#ifdef PINN_EMU
    if (step <= 0) emu::launch(1, 64, 0, [&] { pinn_tick_kernel(step_ptr); });
    emu::launch(blocks, 256, 0, [&] { pinn_adam_kernel(params, grads, exp_avg, exp_avg_sq, mask, n, step_ptr, step, lr, step_size, bc2_sqrt, beta1, beta2, eps, loss_out, off_loss); });
#else
    if (step <= 0) hipLaunchKernelGGL(pinn_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_ptr);
    hipLaunchKernelGGL(pinn_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, mask, (long long)n, (int*)step_ptr, (int)step, lr, step_size, bc2_sqrt, beta1, beta2, eps,
                       loss_out, off_loss);
    if (hipGetLastError() != hipSuccess) return fail("adam kernel launch failed");
#endif
    return 0;
}

int pinn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* mask, int64_t n,
                   int32_t* step_ptr, float lr, float beta1, float beta2, float eps, void* stream) {
    return adam_launch(params, grads, exp_avg, exp_avg_sq, mask, n, step_ptr, 0, lr, beta1, beta2, eps, stream);
}

int pinn_adam_step_at(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* mask, int64_t n,
                      int32_t* step_ptr, int32_t step, float lr, float beta1, float beta2, float eps, float* loss_out,
                      int32_t off_loss, void* stream) {
    if (step < 1) return fail("step must be >= 1");
    if (loss_out && (off_loss < 0 || off_loss >= n)) return fail("off_loss=%d outside the gradient buffer", off_loss);
    return adam_launch(params, grads, exp_avg, exp_avg_sq, mask, n, step_ptr, step, lr, beta1, beta2, eps, stream, loss_out,
                       off_loss);
}

}  // extern "C"
