""" Large batches on every kernel family OUTSIDE the five BASELINE shapes (VERDICT r3 "what's weak" 1): narrow nets (1-2 waves per
workgroup, several workgroups per CU), the breadth kernels (skips, Sin), residual programs, the generic step path, a width that pads
to the streamed-weight-gradient kernels, a third-order shape -- each at 131 072 points, i.e. on a grid that fills every CU with as
many workgroups as the launcher plans, against the oracle evaluated in chunks, and four times over with bit-identical gradients.
Every case records the workgroups per CU of its launch (pinn_last_launch_info) and the narrow ones assert that it was more than one;
the same step capped to ONE workgroup per CU (pinn_debug_max_wgs_per_cu) must agree with the uncapped one to fp32 summation noise. """
import ctypes

import numpy as np
import pytest
import torch

import pinn_configs as pc
from helpers import export_grads, load_params

pytestmark = pytest.mark.gpu

N_BIG = 131072
PI = float(np.pi)


@pytest.fixture(scope='module')
def pa():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    import pydens_amd
    from pydens_amd import engine
    assert engine.load_library().pinn_backend() == b'hip-gfx950'
    return pydens_amd


def launch_info(solver):
    info = (ctypes.c_int32 * 4)()
    assert solver.model.net.lib.pinn_last_launch_info(info) == 0
    return dict(grid=info[0], per_cu=info[1], threads=info[2], lds=info[3],
                kernel=solver.model.net.lib.pinn_last_kernel_name().decode())


def _problem(name, D, V):
    """ (equation, solver kwargs, oracle dtype, path, expects several workgroups per CU) """
    if name in ('skip128', 'skip256', 'sin64', 'sin128', 'program', 'generic', 'burgers64', 'heat64'):
        cfg = pc.make_config(name, D, torch, V=V)
        return cfg['equation'], cfg['solver_kwargs'], torch.float32, ('generic' if name == 'generic' else 'fused'), False
    if name == 'w16_program':         # one wave per workgroup; Burgers: residual program
        eq = lambda f, x, t: D(f, t) - 0.1 * D(D(f, x), x) + f * D(f, x)
        kw = dict(ndims=2, boundary_condition=0, initial_condition=lambda x: torch.sin(PI * x), layout='fa fa fa f',
                  features=[16, 16, 16, 1], activation='Tanh')
        return eq, kw, torch.float32, 'fused', True
    if name == 'w32_affine':          # two waves per workgroup; Poisson with a Sigmoid net: affine residual, generic depth
        eq = lambda f, x, y: D(D(f, x), x) + D(D(f, y), y) - 5 * torch.sin(PI * (x + y))
        kw = dict(ndims=2, boundary_condition=1, layout='fa fa fa f', features=[32, 32, 32, 1], activation='Sigmoid')
        return eq, kw, torch.float32, 'fused', True
    if name == 'w32_generic':         # the same net through pinn_jet_forward -> torch -> pinn_jet_backward
        eq = lambda f, x, y: D(D(f, x), x) + D(D(f, y), y) - 5 * torch.sin(PI * (x + y))
        kw = dict(ndims=2, boundary_condition=1, layout='fa fa fa f', features=[32, 32, 32, 1], activation='Sigmoid')
        return eq, kw, torch.float32, 'generic', True
    if name == 'w100_heat':           # pads to 128: WGX tile kernel + streamed weight-gradient kernel, heat equation with IC + BC
        eq = lambda f, x, y, t: D(D(f, x), x) + D(D(f, y), y) - D(f, t)
        kw = dict(ndims=3, boundary_condition=0, initial_condition=lambda x, y: 10 * x * y * (1 - x) * (1 - y), layout='fa fa fa f',
                  features=[100, 100, 100, 1], activation='Tanh')
        return eq, kw, torch.float32, 'fused', False
    if name == 'w32_third_order':     # dispersive wave: third-order jets (arbitrated in fp64: three nested fp32 sweeps are noisy)
        eq = lambda f, x, t: D(f, t) + f * D(f, x) + 0.1 * D(D(D(f, x), x), x)
        kw = dict(ndims=2, boundary_condition=0.0, initial_condition=lambda x: torch.sin(3.0 * x) * x * x, layout='fa fa fa f',
                  features=[32, 32, 32, 1], activation='Tanh')
        return eq, kw, torch.float64, 'fused', True
    raise KeyError(name)


CASES = ['w16_program', 'w32_affine', 'w32_generic', 'w32_third_order', 'w100_heat', 'skip128', 'skip256', 'sin64', 'program', 'generic',
         'burgers64', 'heat64',          # (round 6: the (x, t) evolution shape on the two-team kernels)
         'sin128']                       # (round 6: the static-activation Sin kernel of the streamed widths)


def run_case(pa, name, n_points, solver_kwargs=None, on_device=True):
    """ the body of the test; tests/test_emu_engine.py runs it on the emulator with a few points (same code path, one workgroup) """
    from oracle import pinn_oracle as po
    torch.manual_seed(CASES.index(name) + 40)
    eq_o, kw, _, path, several = _problem(name, po.D, po.V)
    oracle32 = po.OracleSolver(eq_o, **kw)
    start = oracle32.export_params()
    oracle = po.OracleSolver(eq_o, dtype=torch.float64, **kw)      # the arbiter
    oracle.import_params(start)
    eq_p, kw, _, _, _ = _problem(name, pa.D, pa.V)
    solver = pa.Solver(eq_p, **kw, **(solver_kwargs or {}))
    load_params(solver, start)
    d = kw['ndims']
    pts = np.random.RandomState(77).rand(n_points, d).astype(np.float32)
    ev = oracle.evaluate(pts, chunk=16384)
    g_want = oracle.export_grads()
    ev32, g32 = oracle32.evaluate(pts, chunk=16384), oracle32.export_grads()
    if path == 'generic':
        solver.program = None
    else:
        assert solver.program is not None, solver.program_error
    xs = torch.from_numpy(pts).to(solver.model.flat.device)
    lib = solver.model.net.lib
    lay = solver.model.net.layout

    def step():
        solver.grads.zero_()
        if path == 'fused':
            solver._fused_step(xs, 1)
        else:
            solver._generic_step(xs, ('equation',), (), torch.nn.MSELoss(), 1)
        if on_device:
            torch.cuda.synchronize()
        return solver.grads[:lay.p_total].clone()

    first = step()
    info = launch_info(solver)
    print(f'{name}: {info}')
    if on_device:
        if several:
            assert info['per_cu'] > 1, info            # the case this file exists for: several workgroups share every CU
        assert info['grid'] >= info['per_cu'] * 64, info
    # parity with the oracle at the full batch. A sum over 131 072 points in fp32 is itself good to ~1e-5 (the reference's own chunked
    # fp32 gradients sit up to 3e-5 off their fp64 values on these problems), so the fp64 oracle arbitrates as SURVEY 8c item 5 says:
    # |ours - f64| <= max(2 |ref32 - f64|, 1e-5 |f64|) -- in practice the kernels (per-lane, per-workgroup, then tree sums) are the
    # closer of the two
    loss = float(first[lay.off_loss])
    assert abs(loss - ev['loss']) <= max(2 * abs(ev32['loss'] - ev['loss']), 1e-5 * abs(ev['loss'])), (loss, ev['loss'], ev32['loss'])
    margins = []
    for got, want, w32 in zip(export_grads(solver), g_want, g32):
        if want is not None:
            want = np.asarray(want, dtype=np.float64)
            err = float(np.linalg.norm(np.asarray(got, dtype=np.float64) - want))
            ref_err = float(np.linalg.norm(np.asarray(w32, dtype=np.float64) - want))
            scale = float(np.linalg.norm(want))
            margins.append((err / max(scale, 1e-30), ref_err / max(scale, 1e-30)))
            assert err <= max(2 * ref_err, 1e-5 * scale + 1e-9 * np.sqrt(want.size)), (name, err / scale, ref_err / scale)
    print(f'{name}: worst gradient error vs fp64 {max(m[0] for m in margins):.2e} (the fp32 reference: {max(m[1] for m in margins):.2e})')
    # bitwise repeatability: same inputs, same grid -> the same bits, four times
    for _ in range(3):
        again = step()
        assert torch.equal(first, again), f'{name}: run-to-run different gradients on {info}'
    # ... and the same step on ONE workgroup per CU: another summation order, nothing more
    before = lib.pinn_debug_max_wgs_per_cu(solver.model.net.handle, 1)
    try:
        # (the workspace was sized for the larger grid: partial rows and slabs of a smaller one fit)
        one = step()
        assert launch_info(solver)['per_cu'] == 1
    finally:
        lib.pinn_debug_max_wgs_per_cu(solver.model.net.handle, before)
    a, b = first[:lay.p_core].double().cpu().numpy(), one[:lay.p_core].double().cpu().numpy()
    assert np.linalg.norm(a - b) <= 2e-6 * np.linalg.norm(b), (np.linalg.norm(a - b), np.linalg.norm(b))
    return info


@pytest.mark.parametrize('name', CASES)
def test_large_batch_on_a_full_grid(pa, name):
    # (skip256: 32 768 points = 8 tiles per workgroup of a full grid -- the two CPU oracles of a 5 x 256 net take minutes beyond that)
    run_case(pa, name, N_BIG // 4 if name == 'skip256' else N_BIG)
