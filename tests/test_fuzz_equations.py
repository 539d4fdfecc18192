""" Random residual equations end to end: expression trees over u, u_x, u_t, u_xx, the coordinates and constants
(arithmetic, sin / cos / tanh / sigmoid / abs / exp, squares and cubes), each handed to the product and to the oracle as a
pydens equation callable under an IC + BC ansatz; two Adam iterations must agree (losses, every parameter). Exercises the
tracer, both residual kinds, the in-kernel interpreter with its reverse sweep and the pre-pass on shapes nobody wrote by
hand. CPU: emulated kernels; -m gpu: the HIP library. """
import numpy as np
import pytest
import torch

from conftest import rel_l2
from helpers import FixedBatches, export_params, load_params

LEAVES = ['u', 'ux', 'ut', 'uxx', 'x', 't', 'c']
UNARY = ['sin', 'cos', 'tanh', 'neg', 'sq', 'cube', 'sigmoid', 'abs', 'exps']
BINARY = ['add', 'sub', 'mul', 'divc', 'mulc']


def _gen(rng, depth):
    if depth == 0 or rng.rand() < 0.25:
        leaf = LEAVES[rng.randint(len(LEAVES))]
        return ('c', float(np.round(rng.uniform(-2, 2), 3))) if leaf == 'c' else (leaf,)
    if rng.rand() < 0.4:
        return (UNARY[rng.randint(len(UNARY))], _gen(rng, depth - 1))
    op = BINARY[rng.randint(len(BINARY))]
    if op in ('divc', 'mulc'):
        return (op, _gen(rng, depth - 1), float(np.round(rng.uniform(0.5, 3), 3)))
    return (op, _gen(rng, depth - 1), _gen(rng, depth - 1))


def _ev(tree, env):
    kind = tree[0]
    if kind == 'c':
        return tree[1]
    if kind in env:
        return env[kind]
    a = _ev(tree[1], env)
    if kind in UNARY:
        if isinstance(a, float):
            a = torch.tensor(a)
        if kind == 'neg':
            return -a
        if kind == 'sq':
            return a ** 2
        if kind == 'cube':
            return a * a * a
        if kind == 'exps':
            return torch.exp(0.3 * torch.tanh(a))               # bounded exponent: no overflow on random inputs
        return getattr(torch, kind)(a)
    if kind == 'divc':
        return a / tree[2]
    if kind == 'mulc':
        return tree[2] * a
    b = _ev(tree[2], env)
    return {'add': a + b, 'sub': a - b, 'mul': a * b}[kind]


def _uses(tree, name):
    return tree[0] == name or any(isinstance(c, tuple) and _uses(c, name) for c in tree[1:])


def _equation(tree, D):
    def equation(u, x, t):
        env = {'u': u, 'x': x, 't': t}
        if _uses(tree, 'ux') or _uses(tree, 'uxx'):
            env['ux'] = D(u, x)
        if _uses(tree, 'uxx'):
            env['uxx'] = D(env['ux'], x)
        if _uses(tree, 'ut'):
            env['ut'] = D(u, t)
        return _ev(tree, env) + 0.0 * u + 0.37                  # keeps the field in and the residual away from zero
    return equation


def _run(pa, extra, n_trees, batch):
    from oracle import pinn_oracle as po
    rng = np.random.RandomState(1)
    kw = dict(ndims=2, initial_condition=lambda x: torch.sin(np.pi * x), boundary_condition=0.0, layout='fafaf',
              features=[16, 16, 1], activation='Tanh')
    kinds = {'program': 0, 'affine': 0}
    for trial in range(n_trees):
        tree = _gen(rng, 3)
        if not any(_uses(tree, name) for name in ('u', 'ux', 'ut', 'uxx')):
            continue
        torch.manual_seed(trial)
        oracle = po.OracleSolver(_equation(tree, po.D), **kw)
        solver = pa.Solver(_equation(tree, pa.D), **kw, **extra)
        load_params(solver, oracle.export_params())
        pts = np.random.RandomState(trial).rand(2, batch, 2).astype(np.float32)
        oracle.fit(niters=2, batch_size=batch, points=pts, lr=0.01)
        solver.fit(niters=2, batch_size=batch, sampler=FixedBatches(pts), lr=0.01)
        want = np.array([float(v) for v in oracle.losses])
        if not np.all(np.isfinite(want)):
            continue
        assert solver.last_fit_path == 'fused', (tree, solver.program_error)
        np.testing.assert_allclose([float(v) for v in solver.losses], want, rtol=5e-5, err_msg=str(tree))
        for got, ref in zip(export_params(solver), oracle.export_params()):
            assert rel_l2(got, ref) < 2e-4, tree
        kinds['program' if solver.residual_plan.kind == 0 else 'affine'] += 1
    assert kinds['program'] >= 5 and kinds['affine'] >= 5, kinds


def test_random_equations_on_the_emulated_kernels():
    import ctypes
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
    import build_emu
    import pydens_amd as pa
    from pydens_amd import engine
    _run(pa, dict(lib=engine.bind(ctypes.CDLL(build_emu.build())), device='cpu'), n_trees=40, batch=23)


@pytest.mark.gpu
def test_random_equations_on_the_gpu():
    import pydens_amd as pa
    _run(pa, {}, n_trees=40, batch=523)
