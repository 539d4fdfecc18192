""" shared test helpers: build a pydens_amd Solver for a golden fixture and load the fixture's parameters. """
import numpy as np
import torch

import pinn_configs as pc


class FixedBatches:
    """ sampler plug-in that replays pre-drawn batches (reference model_torch.py:433 call convention). """
    def __init__(self, batches):
        self.batches, self.i = batches, 0

    def sample(self, size):
        out = self.batches[self.i]
        assert out.shape[0] == size
        self.i += 1
        return out.astype(np.float64)


def linear_modules(solver):
    return [m for m in solver.model.conv_block]


def load_params(solver, params):
    with torch.no_grad():
        it = iter(params)
        for lin in linear_modules(solver):
            lin.weight.copy_(torch.as_tensor(next(it)))
            lin.bias.copy_(torch.as_tensor(next(it)))
        solver.model.log_scale.copy_(torch.as_tensor(float(next(it))))


def export_params(solver):
    out = []
    for lin in linear_modules(solver):
        out += [lin.weight.detach().cpu().numpy().copy(), lin.bias.detach().cpu().numpy().copy()]
    out.append(solver.model.log_scale.detach().cpu().numpy().copy())
    return out


def export_grads(solver):
    """ the flat gradient buffer seen through the parameter views (same order as export_params). """
    net, lay = solver.model.net, solver.model.net.layout
    out = []
    for w, b in net.param_views(solver.grads):
        out += [w.detach().cpu().numpy().copy(), b.detach().cpu().numpy().copy()]
    out.append(solver.grads[lay.off_log_scale].detach().cpu().numpy().copy())
    return out


def make_solver(name, pa, gemm=None, **kwargs):
    """ gemm: 'fp32' / 'bf16x3' (Solver.set_gemm_mode); None keeps the default (fp32, or PYDENS_AMD_GEMM) """
    cfg = pc.make_config(name, pa.D, torch)
    solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], **kwargs)
    if gemm is not None:
        solver.set_gemm_mode(gemm)
    return cfg, solver


def ran_split_kernel(solver):
    """ did the last launch take a split-bf16 instantiation (VAR bit 512 of pinn_tile_kernel<...>)? """
    name = solver.model.net.lib.pinn_last_kernel_name().decode()
    return bool(int(name.rstrip('>').split(',')[-1]) & 512)


def fit_rtol(name):
    """ tolerance of a K-step Adam trajectory against the reference's fp32 golden losses. 2e-5 everywhere, except the
    `mixed` fixture: there the REFERENCE's own fp32 trajectory sits 2.4e-5 off the fp64 trajectory at step 3
    (0.07043399 vs 0.07043566; Adam's g/sqrt(g^2) turns rounding of near-zero gradients into +-lr moves) while the
    kernels stay within 5e-6 of fp64 -- so the bound has to cover the fixture's noise, not ours. """
    return 4e-5 if name == 'mixed' else 2e-5


# the gradient of every parameter tensor within 1e-5 (relative L2) of the reference's. Measured on MI355X against
# the reference-generated goldens: 2e-7 .. 6.4e-6 (tools/grad_margins.py, profiles/r03_grad_margins.txt). The absolute floor only
# matters for tensors whose gradient is (nearly) zero.
GRAD_RTOL = 1e-5


def grad_close(got, want, rtol=GRAD_RTOL, atol=1e-9):
    a, b = np.asarray(got, dtype=np.float64).ravel(), np.asarray(want, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b)) <= rtol * float(np.linalg.norm(b)) + atol * np.sqrt(a.size)


# ---- sweeps: the survey's bar everywhere, the fp64 arbiter where fp32 noise of the REFERENCE exceeds it -------------
MARGINS = []            # (test, case, quantity, achieved, bound, arbitrated): written to gpurun_out/grad_margins.txt at session end


def record_margin(test, case, quantity, achieved, bound, arbitrated=False):
    MARGINS.append((test, str(case), quantity, float(achieved), float(bound), bool(arbitrated)))


def close_or_arbitrated(got, want32, want64_fn, rtol, atol=1e-9, k=2.0, note=None, adam_move=None):
    """ |got - ref32| <= rtol |ref32| (+ floor) -- or, where the reference's own fp32 arithmetic is the noisy side, the survey's arbiter:
    |got - f64| <= max(k |ref32 - f64|, rtol |f64| + floor). `want64_fn()` is only evaluated when the first test fails. Returns
    (ok, achieved relative error, arbitrated).
    adam_move (PARAMETERS after a few Adam steps only; = lr * steps, the farthest Adam can move an entry): Adam's update is
    g / sqrt(g^2)-like, so an entry whose gradient sits at the fp32 noise level of the batch sum (a cancelling sum near zero) turns that
    noise into a move of a fraction of lr -- in the fp32 reference and in the kernels alike, in different entries and by different amounts
    (round 6, `fa faR f+a fa f` GELU/Sin/Sigmoid/GELU on the Poisson problem: ONE of 1 287 entries of W2 2.2e-4 off the fp64
    trajectory in the kernels' run, the reference's worst entry 6.6e-5 off). The k = 2 rule on the L2 norm then compares two draws of
    one heavy-tailed noise. So in the arbitrated branch the 1 + n / 512 entries farthest from fp64 are set aside on BOTH sides, each of
    them bounded by 2 % of adam_move, and the rule is applied to the rest. """
    a = np.asarray(got, dtype=np.float64).ravel()
    b = np.asarray(want32, dtype=np.float64).ravel()
    floor = atol * np.sqrt(a.size)
    err, scale = float(np.linalg.norm(a - b)), float(np.linalg.norm(b))
    if err <= rtol * scale + floor:
        return True, err / max(scale, 1e-30), False
    c = np.asarray(want64_fn(), dtype=np.float64).ravel()
    ea, eb = np.abs(a - c), np.abs(b - c)
    outliers_ok = True
    if adam_move is not None and a.size >= 64:
        n_out = 1 + a.size // 512
        ia, ib = np.argsort(ea)[-n_out:], np.argsort(eb)[-n_out:]
        outliers_ok = float(ea[ia].max()) <= 0.02 * adam_move
        ea, eb = np.delete(ea, ia), np.delete(eb, ib)
    err64, ref64, scale64 = float(np.linalg.norm(ea)), float(np.linalg.norm(eb)), float(np.linalg.norm(c))
    return outliers_ok and err64 <= max(k * ref64, rtol * scale64 + floor), err64 / max(scale64, 1e-30), True
