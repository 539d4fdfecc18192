/* pinn.h -- C-ABI of the MI355X-native PINN step engine (libpinn_hip.so).
 *
 * Drop-in boundary for the hot path of pydens `Solver.fit` / `Solver.predict`
 * (reference pydens/model_torch.py:426-464 and :466-487).  The reference has no FFI of its own: every
 * FLOP of its step is an ATen CPU op driven from Python.  Each entry point below names the reference
 * lines whose arithmetic it replaces; the Python host (`pydens_amd`) binds them with ctypes
 * (`pydens_amd/engine.py`), INTEGRATION.md shows the stub a pydens maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (torch tensors
 *     in the Python host); the library allocates nothing on the step path and never synchronises;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value 0 = ok, non-zero = error, text via pinn_last_error(); nothing throws across the ABI;
 *   - all arithmetic is IEEE fp32 (the reference computes in fp32: model_torch.py:35,113,167,353-359,433).
 *
 * Parameter buffer ("flat padded layout", all fp32, described by pinn_layout()):
 *     W1 [HP][d]  b1 [HP]  { Wh_l [HP][HP]  bh_l [HP] } l=0..LH-1   WL [HP]  bL  log_scale  loss_slot  pad
 *     followed by PINN_EXTRA_SLOTS user slots (trainable V(...) scalars, model_torch.py:180-188).
 *   HP = largest hidden width rounded up to 16, 32, 64, 128 or 256 (MFMA tiles; wider nets are rejected by pinn_create),
 *   LH = number of hidden->hidden layers (<= PINN_MAX_LAYERS - 2 = 30).
 *   nn.Linear's [out,in] row-major weight of every layer is the top-left block of its padded matrix, so the
 *   host exposes per-layer nn.Parameter views into this one buffer; padded entries stay zero (Adam mask).
 *   Gradient buffers use the same layout; `loss_slot` receives sum(r^2)/N of the step.
 *
 * Derivative streams (what D(...) needs, model_torch.py:174-178): stream 0 = u; 1..nd = first derivative along
 * direction k; 1+nd..nd+n2 = second derivative along the first n2 directions. A direction dir_cols[k] is an input
 * column c (value c) or a diagonal e_a + e_b / e_a - e_b of two columns (value a | (b + 1) << 4, | PINN_DIR_MINUS for the minus
 * diagonal): the host obtains mixed partials by polarisation, u_ab = (u_vv - u_aa - u_bb) / 2, and (round 5) mixed THIRD-order
 * partials from third derivatives along both diagonals, u_aab = (D3_{a+b} - D3_{a-b} - 2 u_bbb) / 6.
 * Stream arrays are stream-major: [S][N], S = 1 + nd + n2.
 * Third order: wherever an entry point takes `n2`, the value may be PACKED as n2 | n3 << 3 -- n3 of the n2 second-order
 * directions (the first ones; columns or diagonals) also carry a third derivative, streams 1+nd+n2 .. nd+n2+n3, S = 1 + nd + n2 + n3
 * (built: one third-order direction with nd <= 2, i.e. u_xxx-type equations such as KdV; plain n2 < 8 means n3 = 0).
 * Fourth order (round 5): n2 | n3 << 3 | n4 << 6 -- n4 of the n3 third-order directions (the first ones) also carry a fourth derivative,
 * streams behind the third-order ones, S = 1 + nd + n2 + n3 + n4 (built: ONE such direction alone, packed value 73: u, u', u'', u''', u''''
 * along a column or a diagonal -- beam, Kuramoto-Sivashinsky and biharmonic operators; the host assembles
 * u_aabb = (D4_{a+b} + D4_{a-b} - 2 u_aaaa - 2 u_bbbb) / 12).
 * Round 6: WEIGHTED diagonals 2 e_a + e_b / 2 e_a - e_b (| PINN_DIR_DOUBLE): the part of D4 along alpha e_a + e_b that is odd in b is
 * 8 alpha^3 u_aaab + 8 alpha u_abbb, so with A = D4_{a+b} - D4_{a-b} and B = D4_{2a+b} - D4_{2a-b} the host assembles
 * u_aaab = (B - 2 A) / 48 and u_abbb = (8 A - B) / 48 (model_torch.py:174-178 nests D in any order).
 * THREE-column directions e_a +- e_b +- e_c (| (c + 1) << 10, | PINN_DIR_MINUS_C): the third derivatives along the four sign pairs give the
 * partial of three different columns, u_abc = [D3_{+,+} - D3_{+,-} - D3_{-,+} + D3_{-,-}] / 24 (up to third order along such a direction).
 */
#ifndef PINN_H
#define PINN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PINN_MAX_LAYERS   32   /* linear layers */
#define PINN_MAX_INPUTS   8    /* ndims + nparams */
#define PINN_MAX_DIRS     4    /* differentiation directions of one call (3-D space + time) */
#define PINN_EXTRA_SLOTS  16
#define PINN_MAX_OPS      64   /* residual program length */
#define PINN_MAX_CONSTS   32
#define PINN_MAX_REGS     40   /* residual program registers (inputs included) */
#define PINN_MAX_STREAMS  7    /* 1 + nd + n2 */
#define PINN_MAX_AUX      8    /* per-point rows produced by the x-only pre-pass */
#define PINN_MAX_VARS     8    /* trainable V(...) scalars a residual program may read (user slots 0..n_vars-1) */

#define PINN_DIR_MINUS    0x100  /* ORed into a diagonal's direction code: e_a - e_b instead of e_a + e_b */
#define PINN_DIR_DOUBLE   0x200  /* ... : the FIRST column counts twice, 2 e_a + e_b / 2 e_a - e_b (round 6: u_aaab, u_abbb) */
#define PINN_DIR_MINUS_C  0x4000 /* a THIRD column c rides in bits 10..13 as (c + 1) << 10: e_a +- e_b + e_c, with this bit - e_c (round 6: u_abc) */
#define PINN_MAX_SKIPS    4    /* skip connections ('R ... +') per net */
#define PINN_SKIP_PRE     0x100 /* ORed into skip_dst[k] / skip_src[k]: the skip ends / starts IN FRONT of that activation */

#define PINN_ACT_TANH     0
#define PINN_ACT_SIGMOID  1
#define PINN_ACT_SIN      2
#define PINN_ACT_IDENTITY 3    /* no 'a' between two 'f' of the layout */
#define PINN_ACT_SOFTPLUS 4    /* torch.nn.Softplus (beta 1, threshold 20) */
#define PINN_ACT_SILU     5    /* torch.nn.SiLU (swish) */
#define PINN_ACT_GELU     6    /* torch.nn.GELU (erf form) */
/* round 5 -- the rest of what `getattr(nn, activation)()` commonly names (model_torch.py:159, :164-168), each in its torch DEFAULT form;
 * derivatives to fourth order from the pre-activation like codes 4-6 (full breadth kernels only) */
#define PINN_ACT_RELU        7   /* torch.nn.ReLU: derivative z > 0 ? 1 : 0, higher ones 0 (what autograd returns) */
#define PINN_ACT_LEAKYRELU   8   /* torch.nn.LeakyReLU (negative_slope 0.01) */
#define PINN_ACT_ELU         9   /* torch.nn.ELU (alpha 1) */
#define PINN_ACT_SOFTSIGN   10   /* torch.nn.Softsign: z / (1 + |z|) */
#define PINN_ACT_GELU_TANH  11   /* torch.nn.GELU(approximate='tanh') */
#define PINN_ACT_MISH       12   /* torch.nn.Mish: z tanh(softplus(z)) */
#define PINN_ACT_SELU       13   /* torch.nn.SELU */
#define PINN_ACT_TANHSHRINK 14   /* torch.nn.Tanhshrink: z - tanh z */
#define PINN_ACT_LOGSIGMOID 15   /* torch.nn.LogSigmoid */
#define PINN_ACT_LAST       15

typedef struct pinn_net pinn_t;

typedef struct pinn_layout {
    int hp;             /* padded hidden width */
    int lh;             /* hidden->hidden layers */
    int d;              /* input columns */
    int off_w1, off_b1; /* first layer: W1 row stride = d */
    int off_wh;         /* first hidden block; block l at off_wh + l*hidden_stride; W row stride = hp, bias at +hp*hp */
    int hidden_stride;
    int off_wl, off_bl; /* last layer (out = 1) */
    int off_log_scale, off_loss;
    int p_core;         /* floats up to and including loss_slot + pad (what the kernels reduce) */
    int off_extra;      /* first user slot */
    int p_total;        /* p_core + PINN_EXTRA_SLOTS: length of params / grads / Adam-state buffers */
} pinn_layout_t;

/* Residual program: the user's equation (model_torch.py:447-448 `criterion(equation(u_hat, *xs), 0)`) traced
 * by the host into straight-line code over per-point registers.  Registers 0..S-1 hold the u streams,
 * S..S+d-1 the input columns; instruction i writes register S+d+i... (dst is explicit).  The value of the
 * last instruction is the residual r.  word = op | dst<<8 | a<<16 | b<<24; PINN_OP_CONST reads consts[a]. */
enum pinn_op {
    PINN_OP_CONST = 0, PINN_OP_ADD, PINN_OP_SUB, PINN_OP_MUL, PINN_OP_DIV, PINN_OP_NEG,
    PINN_OP_SIN, PINN_OP_COS, PINN_OP_EXP, PINN_OP_LOG, PINN_OP_TANH, PINN_OP_SQRT,
    PINN_OP_POW,      /* a ** consts[b] */
    PINN_OP_ABS, PINN_OP_SIGMOID, PINN_OP_RECIP, PINN_OP_COPY,
    PINN_OP_STORE     /* pre-pass only: aux row b <- register a */
};

typedef struct pinn_program {
    int n_ops;
    int n_consts;
    uint32_t code[PINN_MAX_OPS];
    float consts[PINN_MAX_CONSTS];
} pinn_program_t;

/* Residual = what `equation(u_hat, *xs)` computes per point (model_torch.py:447), in the form the host tracer
 * (pydens_amd/trace.py) reduced it to:
 *   pre     x-only sub-expressions (source terms, variable coefficients): evaluated for ALL points by a tiny
 *           pre-pass kernel; registers 0..d-1 = input columns, PINN_OP_STORE writes a register to aux row b.
 *   kind PINN_RES_AFFINE   r = sum_s C_s * u_s + F with C_s = coef[s] or aux row coef_row[s] (if >= 0) and
 *                          F = src_const or aux row src_row (if >= 0): every linear PDE; no interpreter in the step.
 *   kind PINN_RES_PROGRAM  general pointwise program over streams, inputs and aux rows (registers
 *                          S+d .. S+d+n_aux-1), interpreted per point with its reverse sweep inside the tile kernel.
 *                          n_vars > 0: registers S+d+n_aux .. +n_vars-1 hold the trainable V(...) scalars of user slots
 *                          0..n_vars-1 (model_torch.py:180-188; inverse problems, tutorial :633-645), read from
 *                          params[off_extra + k]; d(loss)/d(slot k) is returned in grads[off_extra + k] like every other
 *                          parameter gradient. */
#define PINN_RES_PROGRAM 0
#define PINN_RES_AFFINE  1

typedef struct pinn_residual {
    int kind;
    int n_aux;
    pinn_program_t pre;
    pinn_program_t program;
    float coef[PINN_MAX_STREAMS];
    int coef_row[PINN_MAX_STREAMS];
    float src_const;
    int src_row;
    /* combined != 0 (with n2 == 1 in the call): the ONE second-order stream (index 1+nd) is sum_k comb_w[k] d2u/dx_k2 over
     * the nd directions -- what a Laplacian / wave / heat operator needs -- instead of one stream per direction. */
    int combined;
    float comb_w[PINN_MAX_DIRS];
    int n_vars;
    /* 1 + user slot of a TRAINABLE constant initial value (`initial_condition=lambda *a: V('init', ...)`, reference
     * examples notebook cells 80-88), 0 = none: the ansatz then adds params[off_extra + slot] instead of the `ic_const`
     * argument and d(loss)/d(slot) = sum over points of d(loss)/du is added to grads[off_extra + slot]. */
    int ic_var1;
    /* ic_rows != 0: the callable initial condition and its derivative streams come from the pre-pass (`pre`), which the
     * host tracer extended by IC(x), dIC/dx_k, ... differentiated symbolically: stream s of the ansatz adds aux row
     * ic_row[s] (if >= 0) or the constant ic_cst[s] -- in the stream layout of the call (combined second-order stream
     * included). Needs has_ic and ic_streams == NULL. Replaces the per-iteration torch autograd over the IC callable
     * (model_torch.py:124-127 under the nested D(...) of :174-178). */
    int ic_rows;
    int ic_row[PINN_MAX_STREAMS];
    float ic_cst[PINN_MAX_STREAMS];
    /* round 6: the x-only pre-pass `pre` is evaluated in DOUBLE precision -- input columns promoted exactly, constant k read from
     * pre_consts64[k] (pre.consts[k] is its fp32 rounding, kept for inspection), every operation and elementary function in fp64,
     * ONE rounding to fp32 when a STORE writes an aux row. Why: source terms like e*pi*cos(e*pi*x) (README.md:78-79) evaluated in fp32
     * carry a SYSTEMATIC error (fl32(pi) != pi moves the argument of the cosine by 3e-8 relative; at arguments of 15 that is 5e-7 in
     * every point, same sign) which survives the sum over the batch in gradients that are cancelling sums -- BASELINE config 4's
     * d(loss)/d(b_L) came out 1.1 - 1.4e-5 from the fp64 oracle for that reason alone (the fp32 reference: 0.5 - 1e-5, same cause;
     * tools/cfg4_bl_probe.py). The pre-pass is a handful of operations per point, once per step. */
    double pre_consts64[PINN_MAX_CONSTS];
} pinn_residual_t;

/* Descriptor of network + ansatz.  Replaces ConvBlockModel.__init__/TorchModel.__init__ bookkeeping
 * (model_torch.py:19-50, :158-168).  layer_dims[0] = ndims+nparams, layer_dims[n_layers] = 1.
 * dom_lo/dom_hi: per-dimension domain limits (model_torch.py:37-46), length ndims.  has_bc/bc_value and has_ic
 * select the ansatz of model_torch.py:107-128 (ndims_spatial = ndims-1 iff has_ic, :25). */
int pinn_create(const int* layer_dims, int n_layers, int act, int ndims, int nparams,
                int has_bc, int has_ic, const float* dom_lo, const float* dom_hi, float bc_value,
                pinn_t** out);
/* General form of the descriptor: per-layer activations and skip connections of the reference's layout strings
 * ('faR fa fa+ f', model_torch.py:143-156).  acts[a], a = 0 .. n_layers-2, is the activation after hidden layer a
 * (PINN_ACT_*).  Skip k adds the output of activation skip_src[k] to the output of activation skip_dst[k]
 * (0 <= src < dst <= n_layers-2, equal widths; round 5: the intervals may nest or cross -- the reference's 'R' / '+' pair like
 * brackets -- as long as at most one skip starts and at most one ends at an activation; forward passes of such a net need scratch:
 * pinn_jet_forward_ws) -- or, with
 * PINN_SKIP_PRE ORed into skip_dst[k], to the pre-activation of hidden layer dst ('+' between 'f' and 'a': the
 * usual residual block act(W h + skip)); with PINN_SKIP_PRE ORed into skip_src[k] the PRE-activation of layer src is what
 * the skip carries ('R' between 'f' and 'a': pre-activation residual blocks). */
int pinn_create_ex(const int* layer_dims, int n_layers, const int* acts, int n_skips, const int* skip_src,
                   const int* skip_dst, int ndims, int nparams, int has_bc, int has_ic, const float* dom_lo,
                   const float* dom_hi, float bc_value, pinn_t** out);
int pinn_destroy(pinn_t* net);
/* Round 6: the ONE parameter some torch activation modules carry, per activation index a = 0 .. n_layers-2 -- nn.LeakyReLU(negative_slope),
 * nn.ELU(alpha), nn.Softplus(beta) (threshold stays 20) -- for nets built from module INSTANCES configured away from torch's defaults
 * (the reference hands instances through to the block, model_torch.py:150, :164-168). par[a] of an activation that takes none must be 0.
 * Without this call every activation has its torch default (0.01 / 1 / 1). Forward values, all derivative orders and the reverse sweep
 * use it (full breadth kernels; oracle/jet_f64.py act_derivs states the formulas). */
int pinn_set_act_params(pinn_t* net, const float* par, int n);
/* pinn_jet_forward (below) with caller-owned scratch: a net with NESTED skip connections parks the jets of its outer skip in global
 * memory between 'R' and '+', in a value-only forward pass too; `workspace` of pinn_workspace_bytes(net, n_points, nd, n2) bytes
 * (16-byte aligned) covers it. Nets without nested skips ignore the workspace; pinn_jet_forward on a nested net fails with a message. */
int pinn_jet_forward_ws(pinn_t* net, const float* params, const float* xs, int64_t n_points, const int* dir_cols, int nd,
                        int n2, const float* ic_streams, float ic_const, float* streams_out, void* workspace, size_t workspace_bytes,
                        void* stream);
int pinn_layout(const pinn_t* net, pinn_layout_t* out);

/* Bytes of scratch the step/backward entry points need for n_points (per-workgroup partial gradients +
 * activation slab + PINN_MAX_AUX pre-pass rows).  Caller allocates once (torch tensor) and reuses it.
 * Widths >= 128 keep the saved jets and pre-activation gradients of every tile of a pass in HBM for the streamed
 * weight-gradient kernel (44 B per point, hidden layer and unit at 4 streams: 5.9 GB for 131 072 points of a 6 x 256 net);
 * batches beyond a 6.5 GB slab budget run chunk by chunk inside the call. */
size_t pinn_workspace_bytes(const pinn_t* net, int64_t n_points, int nd, int n2);

/* Value + derivative streams of the ansatz-transformed network on given points.
 * Replaces ConvBlockModel.forward (model_torch.py:170-172) + anzatc (:107-128) and, for nd>0, the nested
 * autograd sweeps of D (:174-178).  xs [N][d] row-major; ic_streams [S][N] or NULL (then ic_const is the
 * constant initial condition, :31-35); streams_out [S][N].  nd = n2 = 0 is `Solver.predict` (:482-487). */
int pinn_jet_forward(pinn_t* net, const float* params, const float* xs, int64_t n_points,
                     const int* dir_cols, int nd, int n2, const float* ic_streams, float ic_const,
                     float* streams_out, void* stream);

/* Parameter gradient for given upstream stream gradients: grads[p_core] (+)= d(sum_{s,n} grad_streams*streams)/dparams.
 * Replaces `loss.backward()` (model_torch.py:460) for equations the host evaluates itself (generic path) and for
 * constraint terms (:451-457).  The forward jets are recomputed inside the kernel (no state kept between
 * pinn_jet_forward and this call).  accumulate != 0 adds into grads. */
int pinn_jet_backward(pinn_t* net, const float* params, const float* xs, int64_t n_points,
                      const int* dir_cols, int nd, int n2, const float* ic_streams, float ic_const,
                      const float* grad_streams, float* grads, int accumulate, void* workspace, size_t workspace_bytes,
                      void* stream);

/* One fused residual + gradient evaluation: forward jets, ansatz, residual program, mean-square loss and the
 * full reverse sweep, in one launch (the x-only pre-pass runs in its prologue) + a reduction launch.  Replaces model_torch.py:437-460
 * (forward, equation, MSELoss vs zeros, backward).  grads[0..p_core) receives d(loss)/dparams with
 * loss = inv_n_global * sum r^2 over THIS call's points (data-parallel ranks pass 1/N_global and all-reduce
 * the buffer); grads[off_loss] receives this call's share of the loss. */
int pinn_residual_step(pinn_t* net, const pinn_residual_t* residual, const float* params, const float* xs,
                       int64_t n_points, const int* dir_cols, int nd, int n2, const float* ic_streams,
                       float ic_const, float inv_n_global, float* grads, void* workspace, size_t workspace_bytes,
                       void* stream);
/* Same, ADDING gradient and loss to what `grads` already holds: the further terms of a summed loss
 * (model_torch.py:441-457: `loss += criterion(constraint(...), 0)` -- a constraint term is a residual program over the
 * value stream alone, evaluated on the few fixed points the constraint names, nd = n2 = 0). */
int pinn_residual_step_add(pinn_t* net, const pinn_residual_t* residual, const float* params, const float* xs,
                           int64_t n_points, const int* dir_cols, int nd, int n2, const float* ic_streams,
                           float ic_const, float inv_n_global, float* grads, void* workspace, size_t workspace_bytes,
                           void* stream);

/* Single-rank form of one whole `fit` iteration (model_torch.py:437-461): pinn_residual_step with inv_n = 1/n_points and
 * the Adam update of pinn_adam_step fused into the gradient-reduction launch (no all-reduce can sit in between, so
 * data-parallel ranks use the two separate calls). `step` is the 1-based Adam step of THIS update; it is also stored
 * to step_ptr[0] so that the two forms can be mixed. grads still receives the gradient and the loss slot; loss_out
 * (nullable) is one more device address that receives the loss -- the host points it at entry i of its loss history
 * (`self.losses.append(...)`, model_torch.py:464) so that recording an iteration costs no launch and no synchronisation. */
int pinn_residual_adam_step(pinn_t* net, const pinn_residual_t* residual, float* params, const float* xs,
                            int64_t n_points, const int* dir_cols, int nd, int n2, const float* ic_streams,
                            float ic_const, float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* mask,
                            int32_t* step_ptr, int32_t step, float lr, float beta1, float beta2, float eps,
                            float* loss_out, void* workspace, size_t workspace_bytes, void* stream);

/* torch.optim.Adam.step (model_torch.py:461; defaults beta=(0.9,0.999), eps=1e-8, no weight decay) on the flat
 * buffer.  mask[i]==0 freezes entry i (padding, frozen layers/variables: model_torch.py:56-105, :420-421).
 * `step` is the 1-based step count read from device memory (step_ptr[0], int32) so that the call can be
 * replayed from a hipGraph; the kernel increments it. */
int pinn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* mask,
                   int64_t n, int32_t* step_ptr, float lr, float beta1, float beta2, float eps, void* stream);
/* Same update with the 1-based step passed by value (the host counts, as torch.optim does): ONE launch instead of
 * two; the value is also stored to step_ptr[0] so that the forms can be mixed. Data-parallel ranks call this after the
 * gradient all-reduce; loss_out (nullable) then receives grads[off_loss] -- the all-reduced loss of the iteration goes
 * straight into entry i of the host's loss history (`self.losses.append(...)`, model_torch.py:464) in the same launch. */
int pinn_adam_step_at(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* mask,
                      int64_t n, int32_t* step_ptr, int32_t step, float lr, float beta1, float beta2, float eps,
                      float* loss_out, int32_t off_loss, void* stream);

/* The reference's fit loop (model_torch.py:426-464) for the common case -- on-device column sampler, one equation term on the
 * fused path, Adam -- as ONE call that enqueues `k_steps` iterations: iteration k draws batch `call_index0 + k` into `xs`
 * (pinn_sample_points with kind / a / b / seed), runs pinn_residual_adam_step with Adam step `step0 + k` and leaves the
 * iteration's loss in loss_history[k] (device memory). Asynchronous like every entry point; at 100-point batches the host
 * loop is what bounds Solver.fit (three launches per iteration), and this keeps it out of the interpreter. Problems with
 * ic_streams (an initial condition the tracer could not lower) do not take this path. */
int pinn_fit_steps(pinn_t* net, const pinn_residual_t* residual, float* params, float* xs, int64_t n_points,
                   const int* kind, const float* a, const float* b, uint64_t seed, uint64_t call_index0,
                   const int* dir_cols, int nd, int n2, float ic_const, float* grads, float* exp_avg, float* exp_avg_sq,
                   const uint8_t* mask, int32_t* step_ptr, int32_t step0, float lr, float beta1, float beta2, float eps,
                   float* loss_history, int32_t k_steps, void* workspace, size_t workspace_bytes, void* stream);
/* The same chunk as ONE replayable launch graph (hipGraph): for batches of a few thousand points the gaps between the three dependent
 * launches of an iteration are a visible share of it. What changes from iteration to iteration (Philox batch counter, Adam step and its
 * bias corrections, slot of the loss history) is read by the kernels from `ctrl` -- a caller-owned device buffer of at least
 * pinn_fit_ctrl_bytes() bytes, rewritten by an ordinary launch in front of every replay -- indexed by the iteration number baked into the
 * graph's nodes. The first chunk of a configuration runs eagerly and captures; later chunks with the same arguments replay; anything that
 * does not qualify (k_steps != 128 = one whole chunk of Solver.fit, profiling on, capture refused) runs pinn_fit_steps. Same batches, updates and loss history as
 * the eager loop, bit for bit. */
size_t pinn_fit_ctrl_bytes(void);
int pinn_fit_steps_graph(pinn_t* net, const pinn_residual_t* residual, float* params, float* xs, int64_t n_points,
                         const int* kind, const float* a, const float* b, uint64_t seed, uint64_t call_index0,
                         const int* dir_cols, int nd, int n2, float ic_const, float* grads, float* exp_avg, float* exp_avg_sq,
                         const uint8_t* mask, int32_t* step_ptr, int32_t step0, float lr, float beta1, float beta2, float eps,
                         float* loss_history, int32_t k_steps, void* workspace, size_t workspace_bytes, void* ctrl,
                         size_t ctrl_bytes, void* stream);

/* tanh of the hidden layers (round 5). PINN_TANH_FAST (default): sign(z)(1 - t)/(1 + t), t = e^{-2|z|} -- absolute error 1.5e-7, relative
 * accuracy lost to cancellation below |z| ~ 0.3. PINN_TANH_ACCURATE: z P(z^2) below |z| = 0.45 (degree-4 minimax fit, 1.5e-7 RELATIVE); on
 * trained models the gradient error against fp64 drops from 1.7 - 1.9x to 0.6 - 1.2x the fp32 reference's own for +2.5 % kernel time. Built
 * for the Poisson-box shape of BASELINE configs 1 / 2 at width 64 (the other kernels keep the fast form whatever the mode says). */
#define PINN_TANH_FAST     0
#define PINN_TANH_ACCURATE 1
int pinn_set_tanh_mode(pinn_t* net, int mode);
/* Arithmetic of the hidden-layer GEMMs (forward, data gradient, weight gradient) of the fused step -- what ATen's `addmm` /
 * `mm` calls do in the reference (pydens/model_torch.py:170-178, :460), per net:
 *   PINN_GEMM_FP32    (default) v_mfma_f32_16x16x4_f32: exact fp32, bitwise an fmaf chain
 *   PINN_GEMM_BF16X3  every fp32 operand split EXACTLY into three bf16 (hi + mid + lo), the six partial products
 *                     a_i b_j with i + j <= 2 on v_mfma_f32_16x16x32_bf16, fp32 accumulate: the dropped products are below
 *                     2^-24 of |a b| (measured: at or below the rounding error of the fp32 chain). Used by the kernels
 *                     built with it (the BASELINE training shapes: width-64 nets of static depth 3 on the Dirichlet-box /
 *                     ODE-family shapes, widths 128 / 256 of any depth on the heat / wave shapes incl. their streamed
 *                     weight-gradient kernel); every other call keeps the fp32 kernels. pinn_last_kernel_name() tells which one ran.
 * Returns non-zero for an unknown mode. */
#define PINN_GEMM_FP32   0
#define PINN_GEMM_BF16X3 1
int pinn_set_gemm_mode(pinn_t* net, int mode);

/* Measurement hook (bench.py `roofline`): with enable != 0 the step/backward entry points bracket their TILE
 * kernel launch with hipEvents on the launch stream; pinn_last_tile_ms() waits for the last bracket and returns
 * its duration in milliseconds (negative if none). Off by default; never used on the training path. */
int pinn_profile_tile(int enable);
float pinn_last_tile_ms(void);
/* same bracket around the streamed weight-gradient kernel of widths >= 128 (negative if the last step had none) */
float pinn_last_wgrad_ms(void);
/* Instantiation of the tile kernel the last step / forward / backward call launched on this thread's library, e.g.
 * "pinn_tile_kernel<64,2,1,2,3,0,true,16>" (the symbol rocprofv3 shows): bench.py names the kernel it prices with it. */
const char* pinn_last_kernel_name(void);
/* ... and of the streamed weight-gradient kernel of widths >= 128 ("" before the first such launch); the last template argument
 * says whether it was the split-bf16 form (pinn_set_gemm_mode) */
const char* pinn_last_wgrad_kernel_name(void);
/* Geometry of the last tile-kernel launch: out[0] = workgroups of the grid, out[1] = workgroups per CU the grid was planned with
 * (what hipOccupancyMaxActiveBlocksPerMultiprocessor answered for the instantiation, capped -- see pinn_debug_max_wgs_per_cu),
 * out[2] = threads per workgroup, out[3] = bytes of dynamic LDS. The large-batch parity tests assert with it that they really
 * ran several workgroups per CU. Returns non-zero before the first launch. */
int pinn_last_launch_info(int32_t out[4]);

/* Diagnostics (tests, tools/): never used on the training path; they leave results untouched. The three that change how launches
 * are PLANNED act on ONE descriptor (round 5: they were process-wide switches -- two Solvers in a process would have shared them):
 *   pinn_debug_last_kernel        0 = general tile kernel, 2 = shape-specialised tile kernel took the last launch
 *   pinn_debug_prepass_in_kernel  0: x-only pre-pass of this net as its own launch (pinn_aux_kernel) instead of the tile kernel's prologue
 *   pinn_debug_wgx_chunk_bytes    slab budget per pass of the widths >= 128 (default 6.5 GB; <= 0 restores it): tests force
 *                                 multi-chunk steps with a tiny budget; affects pinn_workspace_bytes of this net, so set it first */
int pinn_debug_last_kernel(void);
int pinn_debug_prepass_in_kernel(pinn_t* net, int enable);
int pinn_debug_wgx_chunk_bytes(pinn_t* net, long long bytes);
/*   pinn_debug_max_wgs_per_cu     upper bound of the workgroups per CU this net's tile-kernel grids are planned with (1 .. 4; <= 0
 *                                 restores the default, 4): tests run the same step at 1 and at several workgroups per CU; affects
 *                                 pinn_workspace_bytes (partial rows, slabs), so set it first. Returns the previous bound (-1: null). */
int pinn_debug_max_wgs_per_cu(pinn_t* net, int cap);
/*   pinn_debug_fit_persistent     how pinn_fit_steps_graph runs a chunk of a NARROW net (hidden width <= 32) as ONE launch (pinn_fit_kernel.h):
 *                                 2 (default): the one-CU form -- one hardware workgroup of up to eight virtual workgroups with parameters, Adam
 *                                   state, partial rows and batch resident in LDS; the iteration's only synchronisation is a workgroup
 *                                   barrier -- for batches of at most `rounds` sweeps of those virtual workgroups (pinn_debug_fit_onecu_rounds,
 *                                   default 1: up to 7 - 8 tiles, e.g. BASELINE config 1's 100 points: 13.5 us per iteration against 14.7 us as
 *                                   launch graphs; a second sweep adds ~4.5 us and loses); larger batches replay launch graphs;
 *                                 1: the grid form (at most 64 resident workgroups, one device-scope arrive / wait per iteration). On MI355X
 *                                   that wait crosses the XCDs' L2s (write-back + invalidate per iteration) and costs more than the two launch
 *                                   gaps of a replayed launch graph -- 21 - 34 us against 15 - 19 us per iteration (profiles/r05_small_fit_rate.txt);
 *                                 0: never (launch graphs / eager loop).
 *                                 Returns the previous setting (-1: null).
 *   pinn_debug_fit_onecu_rounds   see above; rounds >= 1, returns the previous value */
int pinn_debug_fit_persistent(pinn_t* net, int mode);
/* Round 6: the grid form (mode 1) waits device-wide once per iteration with a BOUNDED spin; on a timeout every workgroup
 * returns without writing parameters, Adam state or losses back. The launcher now checks that the whole grid is resident before it picks
 * the form, and this call -- it synchronises with the device -- returns 1 (and sets pinn_last_error) if the last grid-form chunk of the
 * calling thread timed out, 0 otherwise, -1 if the flag could not be read. `Solver.fit` calls it behind every grid-form fit and raises. */
int pinn_fit_chunk_status(void);
int pinn_debug_fit_onecu_rounds(pinn_t* net, int rounds);
/*   pinn_debug_fit_graph_stats    out[0] chunks replayed as launch graphs so far, out[1] graphs captured, out[2] captures the runtime
 *                                 refused (those chunks ran eagerly), out[3] the HIP error code of the last refusal */
int pinn_debug_fit_graph_stats(int32_t out[4]);
#ifdef PINN_DEBUG_ABI
/* EXPERIMENT BUILDS ONLY (-DPINN_DEBUG_ABI, tools/variant.sh): the product library neither exports these nor compiles the
 * kernel paths behind them. The flag bits are TIMING experiments -- kernels skip loads / stores / barriers, the results of
 * such a call are meaningless:
 *   pinn_debug_set_flags          bits handed to the kernels (PinnKArgs::debug_flags; 0 = off):
 *                                 2 / 4 slab reads / writes of the widths >= 128 pinned to one tile (L2-resident),
 *                                 8 / 16 / 32 weight-gradient kernel without barrier / LDS staging / HBM loads
 *                                 (-DPINN_CHAIN=1 builds, tools/experiments/pinn_chain_kernel.h: 8 stages consumed but not
 *                                 multiplied, 16 no ring wait, 32 no weight gradient, 64 / 128 no slab reads / writes)
 *   pinn_debug_phase_buffer       device buffer for per-phase cycle counters (-DPINN_PROFILE_PHASES builds, tools/phases.py) */
int pinn_debug_set_flags(int flags);
int pinn_debug_phase_buffer(void* buf);
#endif

/* Collocation points drawn on the device: replaces the host-side sampling of model_torch.py:430-434 (d independent
 * `torch.rand((N,1))` columns, or `sampler.sample(N)` of a NumpySampler product `a & b & ...`, README.md:82) with ONE
 * launch that fills xs [n_points][d] row-major.  Column c is kind[c]: PINN_SAMPLE_UNIFORM a[c] + (b[c] - a[c]) * u,
 * u in [0,1) with 24 random bits; PINN_SAMPLE_NORMAL a[c] + b[c] * z (Box-Muller); PINN_SAMPLE_CONST a[c].
 * kind / a / b are HOST arrays of length d.  Generator: Philox4x32-10 keyed by `seed`, counter = (point index,
 * call_index, column block): the same (seed, call_index) always gives the same batch, any two differ; data-parallel
 * ranks pass different seeds.  oracle/philox.py restates the generator (bit-exact for uniform / constant columns). */
#define PINN_SAMPLE_UNIFORM 0
#define PINN_SAMPLE_NORMAL  1
#define PINN_SAMPLE_CONST   2
int pinn_sample_points(float* xs, int64_t n_points, int d, const int* kind, const float* a, const float* b,
                       uint64_t seed, uint64_t call_index, void* stream);

const char* pinn_last_error(void);
const char* pinn_backend(void);   /* "hip-gfx950" for the product library */

#ifdef __cplusplus
}
#endif
#endif
