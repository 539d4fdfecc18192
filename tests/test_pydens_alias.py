""" The reference's import name (callers do `from pydens import Solver, D, V, ...`, pydens/__init__.py:4-5, README.md:26):
the `pydens` alias package at the repo root, and the five problems of tutorials/1. Solving PDEs.ipynb run as SCRIPTS whose import
lines are the notebook's, untouched (cells 1, 12-16, 19-24, 28-34, 37-42, 50-62; plotting left out). The scripts run in a
subprocess on the HIP library (-m gpu): each trains as the notebook does and checks the notebook's own known answers (the analytic
solutions it plots against; the hard-bound initial / boundary values). """
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HEADER = '''
import sys
import numpy as np
import torch
from torch import nn
from pydens import Solver, D, V, ConvBlockModel
from pydens import NumpySampler as NS
torch.manual_seed(3)

def cart_prod(*arrs):
    grids = np.meshgrid(*arrs, indexing='ij')
    return np.stack(grids, axis=-1).reshape(-1, len(arrs))
'''

SCRIPTS = {
    # cells 12-16: f' = 2 pi cos(2 pi x), f(0) = 1/2 -> sin(2 pi x) + 1/2
    'ode': '''
def ode(f, x):
    return D(f, x) - 2 * np.pi * torch.cos(2 * np.pi * x)

solver = Solver(ode, ndims=1, initial_condition=.5, activation='Tanh', layout='fafaf',
                features=[12, 10, 1])
solver.fit(niters=500, batch_size=400, lr=0.02)
xs = torch.tensor(np.linspace(0, 1, 100)).float()
fs = solver.predict(xs)
err = np.abs(fs[:, 0] - (np.sin(2 * np.pi * xs.numpy()) + .5)).max()
assert fs.shape == (100, 1) and abs(fs[0, 0] - .5) < 1e-6, fs[0]
assert err < 0.15, err
assert len(solver.losses) == 500 and float(solver.losses[-1]) < 0.05 * float(solver.losses[0])
''',
    # cells 19-24: Poisson with boundary value 1
    'poisson': '''
def pde(f, x, y):
    return D(D(f, x), x) + D(D(f, y), y) - 5 * torch.sin(np.pi * (x + y))

solver = Solver(pde, ndims=2, boundary_condition=1,
                model=ConvBlockModel,
                layout='fafaf', features=[10, 10, 1], activation='Tanh')
solver.fit(niters=300, batch_size=400, lr=0.02)
grid = cart_prod(np.linspace(0, 1, 100), np.linspace(0, 1, 100))
approxs = solver.predict(grid[:, 0:1], grid[:, 1:2]).reshape((100, 100))
for edge in (approxs[0], approxs[-1], approxs[:, 0], approxs[:, -1]):
    assert np.abs(edge - 1).max() < 1e-6                     # the ansatz binds the boundary value exactly
assert float(solver.losses[-1]) < 0.05 * float(solver.losses[0]), (solver.losses[0], solver.losses[-1])
''',
    # cells 28-34: parametric family f' = e pi cos(e pi x), f(0) = 2 -> sin(e pi x) + 2 (the notebook trains 7 000 iterations)
    'odeparam': '''
def odeparam(f, x, e):
    return D(f, x) - e * np.pi * torch.cos(e * np.pi * x)

solver = Solver(odeparam, ndims=1, initial_condition=2.0, nparams=1)
sampler = NS('u') & NS('u', low=.5, high=5.5)
solver.fit(niters=3000, batch_size=700, sampler=sampler, lr=0.01)
xs = torch.tensor(np.linspace(0, 1, 100)).float()
eps = 1
approxs = solver.predict(xs, eps)
assert approxs.shape == (100, 1) and abs(approxs[0, 0] - 2) < 1e-6
assert np.abs(approxs[:, 0] - (np.sin(eps * np.pi * xs.numpy()) + 2)).max() < 0.35
assert float(solver.losses[-1]) < 0.2 * float(solver.losses[0])
''',
    # cells 37-42: heat equation family with a callable initial condition and zero boundary
    'heat': '''
def pde(f, x, y, t, a):
    return D(D(f, x), x) + D(D(f, y), y) - a * D(f, t)

solver = Solver(pde, ndims=3, nparams=1,
                initial_condition=lambda x, y: 10 * x * y * (1 - x) * (1 - y),
                boundary_condition=0, layout='fafaf', features=[30, 40, 1], activation='Sigmoid')
sampler = NS('u', dim=2) & NS('u', low=0, high=.5) &  NS('u', low=.1, high=4)
solver.fit(niters=300, batch_size=1500, lr=0.001)

def get_approxs(t=.1, param=1.):
    grid = cart_prod(np.linspace(0, 1, 100),
                     np.linspace(0, 1, 100))
    xs, ys = grid[:, 0:1], grid[:, 1:2]
    return solver.predict(xs, ys, t, param).reshape((100, 100))

at0 = get_approxs(t=0.)
g = cart_prod(np.linspace(0, 1, 100), np.linspace(0, 1, 100))
ic = (10 * g[:, 0] * g[:, 1] * (1 - g[:, 0]) * (1 - g[:, 1])).reshape((100, 100))
assert np.abs(at0 - ic).max() < 1e-5                       # u(x, y, t0) = IC exactly
later = get_approxs(t=.1)
for edge in (later[0], later[-1], later[:, 0], later[:, -1]):
    assert np.abs(edge).max() < 1e-6
assert len(solver.losses) == 300 and np.isfinite(float(solver.losses[-1]))
''',
    # cells 50-62: trainable variable in the equation, then the constraint term
    'odevar': '''
def odevar(f, x):
    return (D(f, x) - 2 * np.pi * torch.cos(2 * np.pi * x)
            + V('new_var', data=torch.Tensor([1.0])))

solver = Solver(odevar, ndims=1, initial_condition=1,
                constraints=lambda f, x: f(torch.tensor([0.5])))
solver.fit(niters=200, batch_size=500, lr=0.1)
xs = torch.Tensor(np.linspace(0, 1, 100))
approxs = solver.predict(xs)
assert abs(approxs[0, 0] - 1) < 1e-6
assert float(solver.losses[-1]) < 0.1 * float(solver.losses[0])
before = abs(float(solver.predict(torch.tensor([0.5]))[0, 0]))
solver.fit(niters=100, batch_size=100, lr=0.1,
           loss_terms=['equation', 'constraint_0'])
approxs = solver.predict(xs)
assert len(solver.losses) == 300 and hasattr(solver.model, 'new_var')
mid = abs(float(solver.predict(torch.tensor([0.5]))[0, 0]))
assert mid < before and mid < 1.0, (before, mid)            # the constraint pulls f(0.5) towards 0 (the notebook's second plot)
''',
}


def test_alias_exports_the_reference_names():
    import pydens
    import pydens.model_torch as mt
    import pydens_amd
    for name in ('Solver', 'D', 'V', 'TorchModel', 'ConvBlockModel', 'NumpySampler'):       # pydens/__init__.py:4-5
        assert getattr(pydens, name) is getattr(pydens_amd, name), name
    assert mt.Solver is pydens.Solver and mt.D is pydens.D and mt.V is pydens.V
    assert pydens.__version__ == '1.0.2'
    # the sampler algebra the notebooks use under this name
    both = pydens.NumpySampler('u') & pydens.NumpySampler('u', low=.5, high=5.5)
    assert both.sample(7).shape == (7, 2)


@pytest.mark.parametrize('name', sorted(SCRIPTS))
def test_tutorial_scripts_parse(name):
    compile(HEADER + SCRIPTS[name], f'<tutorial {name}>', 'exec')


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(SCRIPTS))
def test_tutorial_script_runs_untouched_on_the_gpu(name):
    res = subprocess.run([sys.executable, '-c', HEADER + SCRIPTS[name]], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
