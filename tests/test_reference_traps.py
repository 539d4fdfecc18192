""" The semantic traps of the reference listed in SURVEY.md 8a (numbers below), pinned one by one on the product (emulated
kernels). Those that are pure arithmetic (5: BC transform before IC transform; 8: loss target / broadcast) are pinned by
the golden-fixture and oracle trajectory tests; here: the API-visible ones. """
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))


@pytest.fixture(scope='module')
def kw():
    import build_emu
    from pydens_amd import engine
    return dict(_lib=engine.bind(ctypes.CDLL(build_emu.build())), device='cpu')


def _ode(pa):
    return lambda u, t: pa.D(u, t) - u


def test_variable_order_and_spatial_dimensions(kw):
    import pydens_amd as pa
    # 1: columns are (spatial..., t, params...); 2: ndims_spatial = ndims - 1 iff there is an initial condition
    heat = pa.Solver(lambda u, x, y, t, a: pa.D(u, t) - a * (pa.D(pa.D(u, x), x) + pa.D(pa.D(u, y), y)), ndims=3, nparams=1,
                     initial_condition=lambda x, y: x * y, boundary_condition=0, layout='faf', features=[8, 1],
                     activation='Tanh', **kw)
    assert heat.model.total == 4 and heat.model.ndims_spatial == 2
    ode = pa.Solver(_ode(pa), ndims=1, initial_condition=1.0, layout='faf', features=[8, 1], activation='Tanh', **kw)
    assert ode.model.ndims_spatial == 0
    box = pa.Solver(lambda u, x, y: pa.D(pa.D(u, x), x) + pa.D(pa.D(u, y), y), ndims=2, boundary_condition=1, layout='faf',
                    features=[8, 1], activation='Tanh', **kw)
    assert box.model.ndims_spatial == 2
    # the field at t = t0 is the initial condition, on the box boundary the boundary value (hard constraints of the ansatz)
    assert np.allclose(ode.predict(np.zeros(3)), 1.0, atol=1e-6)
    edge = np.array([0.0, 1.0, 0.3, 0.7])
    assert np.allclose(box.predict(edge, np.array([0.2, 0.9, 0.0, 1.0])), 1.0, atol=1e-6)
    assert np.allclose(heat.predict(0.3, 0.6, 0.0, 2.0), 0.18, atol=1e-6)             # IC(x, y) = x y at t0 = 0


def test_domain_is_replicated_and_the_default_sampler_ignores_it(kw):
    import pydens_amd as pa
    # 3: domain=(lo, hi) applies to every dimension; the default sampler draws U[0,1) whatever the domain (model_torch.py:431)
    solver = pa.Solver(lambda u, x, y: pa.D(pa.D(u, x), x) + pa.D(pa.D(u, y), y), ndims=2, boundary_condition=2.0,
                       domain=(-1, 2), layout='faf', features=[8, 1], activation='Tanh', **kw)
    assert [tuple(d) for d in solver.model.domain] == [(-1, 2), (-1, 2)]
    xs = solver._sample(2000, None).numpy()
    assert xs.min() >= 0.0 and xs.max() < 1.0
    assert np.allclose(solver.predict(np.array([-1.0, 2.0]), np.array([0.5, 0.5])), 2.0, atol=1e-6)


def test_initial_condition_calling_conventions(kw):
    import pydens_amd as pa
    # 4: a callable IC receives 1-D [N] column slices of the SPATIAL columns only; a constant IC is broadcast
    seen = []

    def ic(x):
        seen.append(tuple(x.shape))
        return torch.sin(x)
    solver = pa.Solver(lambda u, x, t: pa.D(u, t) - pa.D(pa.D(u, x), x), ndims=2, initial_condition=ic, boundary_condition=0,
                       layout='faf', features=[8, 1], activation='Tanh', **kw)
    seen.clear()
    solver.predict(np.linspace(0, 1, 7), 0.0)
    assert seen and all(len(shape) == 1 for shape in seen) and seen[-1] == (7,)


def test_log_scale_is_always_a_parameter(kw):
    import pydens_amd as pa
    # 6: log_scale is registered and handed to the optimizer even without an IC; its gradient is then zero / None and
    # Adam leaves it alone (reference: grad None -> skipped)
    box = pa.Solver(lambda u, x, y: pa.D(pa.D(u, x), x) + pa.D(pa.D(u, y), y) - 1.0, ndims=2, boundary_condition=1,
                    layout='faf', features=[8, 1], activation='Tanh', **kw)
    assert 'log_scale' in dict(box.model.named_parameters())
    box.fit(niters=3, batch_size=20, lr=0.1)
    assert float(box.model.log_scale) == 0.0
    ode = pa.Solver(_ode(pa), ndims=1, initial_condition=1.0, layout='faf', features=[8, 1], activation='Tanh', **kw)
    ode.fit(niters=3, batch_size=20, lr=0.1)
    assert float(ode.model.log_scale) != 0.0


def test_every_fit_call_builds_a_fresh_optimizer_unless_told_otherwise(kw):
    import pydens_amd as pa
    # 7: fit(optimizer='Adam') resets moments and step count (model_torch.py:419-422); optimizer=None reuses
    solver = pa.Solver(_ode(pa), ndims=1, initial_condition=1.0, layout='faf', features=[8, 1], activation='Tanh', **kw)
    solver.fit(niters=4, batch_size=16)
    first = solver.optimizer
    assert first.t == 4
    solver.fit(niters=2, batch_size=16)
    assert solver.optimizer is not first and solver.optimizer.t == 2
    second = solver.optimizer
    solver.fit(niters=3, batch_size=16, optimizer=None)
    assert solver.optimizer is second and second.t == 5
    fresh = pa.Solver(_ode(pa), ndims=1, initial_condition=1.0, layout='faf', features=[8, 1], activation='Tanh', **kw)
    with pytest.raises(ValueError):
        fresh.fit(niters=1, batch_size=4, optimizer=None)


def test_types_of_losses_and_predictions(kw):
    import pydens_amd as pa
    # 9: losses are 0-d numpy arrays, one per iteration; predict returns a float32 ndarray [N, 1]; 11: train() / eval()
    solver = pa.Solver(_ode(pa), ndims=1, initial_condition=1.0, layout='faf', features=[8, 1], activation='Tanh', **kw)
    solver.fit(niters=3, batch_size=16)
    assert len(solver.losses) == 3 and all(isinstance(v, np.ndarray) and v.shape == () for v in solver.losses)
    assert solver.model.training
    out = solver.predict(np.linspace(0, 1, 5))
    assert isinstance(out, np.ndarray) and out.shape == (5, 1) and out.dtype == np.float32
    assert not solver.model.training
    # 10: scalars are tiled to the longest argument; an ndarray of another size is replaced by its first element, tiled
    two = pa.Solver(lambda u, x, t: pa.D(u, t) - pa.D(pa.D(u, x), x), ndims=2, initial_condition=lambda x: x, boundary_condition=0,
                    layout='faf', features=[8, 1], activation='Tanh', **kw)
    a = two.predict(np.linspace(0, 1, 4), 0.25)
    b = two.predict(np.linspace(0, 1, 4), np.full(4, 0.25))
    c = two.predict(np.linspace(0, 1, 4), np.array([0.25, 0.9]))             # wrong size: first element, tiled (:355-356)
    assert a.shape == (4, 1) and np.array_equal(a, b) and np.array_equal(a, c)


def test_state_dict_round_trip_through_the_flat_buffer(kw):
    """ checkpoint / resume the torch way: the nn.Parameter views of the flat kernel buffer (layers, log_scale, trainable
    variables) survive model.state_dict() -> load_state_dict() into a fresh Solver bit for bit """
    import pydens_amd as pa

    def make():
        return pa.Solver(lambda u, x: pa.D(u, x) - 2 * np.pi * torch.cos(2 * np.pi * x) + pa.V('new_var', data=torch.Tensor([1.0])),
                         ndims=1, initial_condition=1, layout='fafaf', features=[12, 10, 1], activation='Tanh', **kw)
    first = make()
    first.fit(niters=5, batch_size=32, lr=0.05)
    state = first.model.state_dict()
    assert set(state) == {'log_scale', 'new_var', 'conv_block.fc1.weight', 'conv_block.fc1.bias', 'conv_block.fc2.weight',
                          'conv_block.fc2.bias', 'conv_block.fc3.weight', 'conv_block.fc3.bias'}
    second = make()
    second.model.load_state_dict(state)
    assert torch.equal(first.model.flat, second.model.flat)
    xs = np.linspace(0, 1, 9).astype(np.float32)
    assert np.array_equal(first.predict(xs), second.predict(xs))
    assert float(second.model.new_var.detach()) == float(first.model.new_var.detach()) != 1.0
