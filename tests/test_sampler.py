""" On-device collocation sampler (include/pinn.h pinn_sample_points): the numpy restatement is pinned to the
Random123 known-answer vectors of Philox4x32-10, the kernel is compared with it bit for bit -- through the emulator on
CPU, through the HIP library on the GPU (-m gpu) -- and the host plumbing (default sampler, NumpySampler products) is
checked to take the one-launch path. """
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

from oracle import philox

COLUMNS = [(philox.UNIFORM, 0.0, 1.0), (philox.UNIFORM, 1.0, 5.0), (philox.CONST, 4.0, 0.0), (philox.UNIFORM, -2.5, 0.5),
           (philox.NORMAL, 10.0, 0.1), (philox.UNIFORM, 0.1, 4.0), (philox.NORMAL, 0.0, 1.0), (philox.UNIFORM, 0.0, 0.5)]


def test_oracle_matches_the_random123_known_answers():
    for ctr, key, want in philox.KNOWN_ANSWERS:
        got = philox.philox4x32_10(*[np.uint32(c) for c in ctr], key[0], key[1])
        assert [int(g) for g in got] == list(want)


@pytest.fixture(scope='module')
def emu_net():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
    import build_emu
    from pydens_amd import engine
    lib = engine.bind(ctypes.CDLL(build_emu.build()))
    return engine.Net([2, 16, 1], 'tanh', 2, lib=lib)


def _compare(net, device, n, columns, seed, call):
    xs = torch.full((n, len(columns)), float('nan'), dtype=torch.float32, device=device)
    net.sample_points(xs, columns, seed, call)
    got, want = xs.cpu().numpy(), philox.sample_points(n, columns, seed, call)
    for c, (kind, _, _) in enumerate(columns):
        if kind == philox.NORMAL:
            np.testing.assert_allclose(got[:, c], want[:, c], rtol=0, atol=2e-6 * max(1.0, abs(columns[c][2])) * 6)
        else:
            assert np.array_equal(got[:, c].view(np.uint32), want[:, c].view(np.uint32)), f'column {c}'
    return got


@pytest.mark.parametrize('n', [1, 255, 257, 1000])
def test_emulated_kernel_matches_the_oracle_bit_for_bit(emu_net, n):
    for d in (1, 2, 3, 5, 8):
        _compare(emu_net, 'cpu', n, COLUMNS[:d], seed=0x0123456789abcdef, call=7 + d)
    a = _compare(emu_net, 'cpu', n, COLUMNS[:3], seed=11, call=(1 << 40) + 3)
    b = _compare(emu_net, 'cpu', n, COLUMNS[:3], seed=11, call=(1 << 40) + 4)
    c = _compare(emu_net, 'cpu', n, COLUMNS[:3], seed=12, call=(1 << 40) + 3)
    assert not np.array_equal(a[:, 0], b[:, 0]) and not np.array_equal(a[:, 0], c[:, 0])


def test_distribution_moments(emu_net):
    x = _compare(emu_net, 'cpu', 60000, COLUMNS, seed=5, call=0)
    for c, (kind, a, b) in enumerate(COLUMNS):
        if kind == philox.UNIFORM:
            assert x[:, c].min() >= a and x[:, c].max() < b
            assert abs(x[:, c].mean() - (a + b) / 2) < 0.01 * (b - a)
            assert abs(x[:, c].std() - (b - a) / np.sqrt(12)) < 0.01 * (b - a)
        elif kind == philox.NORMAL:
            assert abs(x[:, c].mean() - a) < 0.02 * b and abs(x[:, c].std() - b) < 0.02 * b
        else:
            assert np.all(x[:, c] == a)
    assert abs(np.corrcoef(x[:, 0], x[:, 1])[0, 1]) < 0.02 and abs(np.corrcoef(x[:-1, 0], x[1:, 0])[0, 1]) < 0.02


def test_solver_draws_default_and_product_samplers_with_the_kernel(emu_net):
    import pydens_amd as pa
    torch.manual_seed(3)
    solver = pa.Solver(lambda u, x, e: pa.D(u, x) - e * torch.cos(e * x), ndims=1, nparams=1, initial_condition=2.0,
                       layout='faf', features=[8, 1], activation='Tanh', _lib=emu_net.lib, device='cpu')
    xs = solver._sample(500, None)
    want = philox.sample_points(500, [(0, 0.0, 1.0)] * 2, solver._sample_seed, 0)
    assert np.array_equal(xs.numpy(), want) and solver._sample_calls == 1
    sampler = pa.NumpySampler('uniform') & pa.NumpySampler('uniform', low=1, high=5)       # reference README.md:82
    xs = solver._sample(500, sampler).numpy()
    assert np.array_equal(xs, philox.sample_points(500, [(0, 0.0, 1.0), (0, 1.0, 5.0)], solver._sample_seed, 1))
    assert xs[:, 1].min() >= 1 and xs[:, 1].max() < 5
    assert (pa.NumpySampler('u', dim=2) & pa.NumpySampler('n', loc=1, scale=2) & 3.0).columns() == \
        [(0, 0.0, 1.0), (0, 0.0, 1.0), (1, 1.0, 2.0), (2, 3.0, 0.0)]
    assert pa.NumpySampler('exponential').columns() is None                                  # numpy path kept
    n0 = len(solver.losses)
    solver.fit(niters=3, batch_size=64, sampler=sampler)
    assert len(solver.losses) == n0 + 3 and solver._sample_calls == 5


@pytest.mark.gpu
def test_gpu_kernel_matches_the_oracle_bit_for_bit():
    from pydens_amd import engine
    net = engine.Net([2, 16, 1], 'tanh', 2)
    for n in (1, 257, 65536, 1 << 20):
        _compare(net, 'cuda', n, COLUMNS[:3], seed=0xfeedfacecafebeef, call=n)
    x = _compare(net, 'cuda', 200000, COLUMNS, seed=1, call=(1 << 33) + 1)
    assert abs(x[:, 0].mean() - 0.5) < 0.005 and abs(x[:, 6].std() - 1.0) < 0.01
