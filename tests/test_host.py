""" Host-side logic that needs no kernels: input casting, samplers, layout parsing, the C-ABI surface. """
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def test_reshape_and_concat_follows_the_reference_rules():
    from pydens_amd import Solver
    from oracle.pinn_oracle import OracleSolver
    cases = [
        (torch.linspace(0, 1, 5), 4),                                  # scalar tiled to the longest argument
        (np.linspace(0, 1, 6).reshape(6, 1), np.array([7.0, 8.0])),    # ndarray of another size -> first element
        ([0.1, 0.2, 0.3], 2.5, np.arange(3.0)),
        (0.5,),
    ]
    oracle_cast = OracleSolver.reshape_and_concat
    for args in cases:
        got = Solver.reshape_and_concat(args)
        want = oracle_cast(type('S', (), {'dtype': torch.float32})(), args)
        assert got.dtype == torch.float32 and got.shape == want.shape
        assert torch.equal(got, want)


def test_samplers():
    from pydens_amd import NumpySampler, NS, ConstantSampler
    s = NumpySampler('uniform', seed=1) & NumpySampler('u', low=1, high=5, seed=2)
    pts = s.sample(1000)
    assert pts.shape == (1000, 2) and pts.dtype == np.float64
    assert 0 <= pts[:, 0].min() and pts[:, 0].max() < 1 and 1 <= pts[:, 1].min() and pts[:, 1].max() < 5
    assert NS('u', dim=2).sample(10).shape == (10, 2)
    t = (NS('u', low=2, high=3) & NS('n', loc=10, scale=0.1) & ConstantSampler(4.0)).sample_device(4000, 'cpu')
    assert t.shape == (4000, 3) and t.dtype == torch.float32
    assert 2 <= float(t[:, 0].min()) and float(t[:, 0].max()) < 3 and abs(float(t[:, 1].mean()) - 10) < 0.02
    assert torch.all(t[:, 2] == 4.0)
    with pytest.raises(ValueError):
        NumpySampler('no_such_distribution')


def test_layout_parsing():
    from pydens_amd.model import parse_fc_layout
    assert parse_fc_layout('fa fa fa f', [10, 12, 15, 1], 'Tanh') == ([10, 12, 15, 1], ['Tanh'] * 3, [])
    assert parse_fc_layout('fafaf', (20, 30, 1), torch.nn.Sigmoid) == ([20, 30, 1], ['Sigmoid'] * 2, [])
    # the reference's docstring example with a skip (model_torch.py:155): output of layer 2 += output of layer 0
    assert parse_fc_layout('faR fa fa+ f', [5, 10, 5, 1], 'Sigmoid') == ([5, 10, 5, 1], ['Sigmoid'] * 3, [(0, 2, False, False)])
    assert parse_fc_layout('fa R fa + R fa + f', [8, 8, 8, 1], ['Tanh', torch.sin, torch.nn.Sigmoid()])[1:] == \
        (['Tanh', 'Sin', 'Sigmoid'], [(0, 1, False, False), (1, 2, False, False)])
    # the usual residual block act(W h + skip): '+' between the dense layer and its activation
    assert parse_fc_layout('faR fa f+a f', [8, 8, 8, 1], 'Tanh')[1:] == (['Tanh'] * 3, [(0, 2, True, False)])
    assert parse_fc_layout('fRa fa f+a f', [8, 8, 8, 1], 'Tanh')[2] == [(0, 2, True, True)]       # pre-activation block: z to z
    assert parse_fc_layout('ff', [5, 1], 'Sigmoid') == ([5, 1], ['Identity'], [])
    assert parse_fc_layout('fa f fa f', [5, 4, 3, 1], 'Tanh')[1] == ['Tanh', 'Identity', 'Tanh']
    with pytest.raises(NotImplementedError):
        parse_fc_layout('fafaf', [5, 10, 3], 'Sigmoid')              # vector-valued output
    with pytest.raises(NotImplementedError):
        parse_fc_layout('fafa', [5, 1], 'Sigmoid')                   # activation on the output
    with pytest.raises(NotImplementedError):
        parse_fc_layout('faaf', [5, 1], 'Sigmoid')
    with pytest.raises(NotImplementedError):
        parse_fc_layout('cafaf', [5, 5, 1], 'Sigmoid')               # conv letters are out of scope
    with pytest.raises(ValueError):
        parse_fc_layout('faR fa+ f', [5, 6, 1], 'Sigmoid')           # skip joins different widths
    with pytest.raises(ValueError):
        parse_fc_layout('fafaf', [5, 5, 1], ['Tanh'])                # too few activations


def test_domain_validation_matches_reference():
    from pydens_amd.model import TorchModel

    class M(TorchModel):
        def forward(self, xs):
            return xs
    assert M(ndims=2, domain=(0, 2)).domain == [(0, 2), (0, 2)]
    assert M(ndims=2, domain=[(0, 1), (1, 2)], initial_condition=1.0).ndims_spatial == 1
    with pytest.raises(ValueError):
        M(ndims=2, domain=5)
    with pytest.raises(ValueError):
        M(ndims=2, domain=['a', 'b'])


def test_no_cpu_fallback_without_device_or_library(monkeypatch):
    import pydens_amd
    from pydens_amd import engine
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            pydens_amd.Solver(lambda f, x: pydens_amd.D(f, x), ndims=1)
    monkeypatch.setattr(engine, '_LIB', None)
    monkeypatch.setattr(engine, 'LIB_NAME', 'libpinn_missing.so')
    with pytest.raises(RuntimeError, match='no fallback'):
        engine.load_library()


def test_hip_library_exports_every_declared_symbol():
    """ include/pinn.h is the contract: the gfx950 build must load and export each entry point (no compute here). """
    from pydens_amd.csrc import build as hip_build
    from pydens_amd import engine
    path = hip_build.build()
    header = open(os.path.join(ROOT, 'include', 'pinn.h')).read()
    # (the experiment-only entry points sit behind PINN_DEBUG_ABI: the product neither declares nor exports them)
    product, debug_only = re.subn(r'#ifdef PINN_DEBUG_ABI.*?#endif', '', header, flags=re.S)
    assert debug_only == 1
    declared = sorted(set(re.findall(r'\b(pinn_[a-z_0-9]+)\s*\(', product)))
    assert set(declared) == set(engine.ABI_SYMBOLS)
    lib = ctypes.CDLL(path)
    for sym in declared:
        assert hasattr(lib, sym), sym
    for sym in ('pinn_debug_set_flags', 'pinn_debug_phase_buffer'):
        assert not hasattr(lib, sym), sym
    engine.bind(lib)
    assert lib.pinn_backend() == b'hip-gfx950'
    lay = engine.Layout()
    handle = ctypes.c_void_p()
    dims = (ctypes.c_int * 6)(2, 64, 64, 64, 64, 1)
    assert lib.pinn_create(dims, 5, 0, 2, 0, 1, 0, None, None, 1.0, ctypes.byref(handle)) == 0
    assert lib.pinn_layout(handle, ctypes.byref(lay)) == 0
    assert (lay.hp, lay.lh, lay.d, lay.p_core) == (64, 3, 2, 64 * 2 + 64 + 3 * (64 * 64 + 64) + 64 + 4)
    lib.pinn_destroy(handle)
    bad = (ctypes.c_int * 3)(2, 600, 1)          # (round 6: widths up to 512 exist)
    assert lib.pinn_create(bad, 2, 0, 2, 0, 0, 0, None, None, 0.0, ctypes.byref(handle)) != 0
    assert b'600' in lib.pinn_last_error()
    wide = (ctypes.c_int * 3)(2, 300, 1)
    assert lib.pinn_create(wide, 2, 0, 2, 0, 0, 0, None, None, 0.0, ctypes.byref(handle)) == 0
    assert lib.pinn_layout(handle, ctypes.byref(lay)) == 0 and lay.hp == 512
    lib.pinn_destroy(handle)


def test_host_constants_of_a_constraint_are_cached_on_the_device_by_content():
    """ Solver._points_on_device: the fixed points a constraint builds in every call (`f(torch.tensor([0.5]))`) are uploaded once and
    found again by content -- what keeps the generic step recordable as a launch graph (a pageable-memory upload is not capturable);
    anything that needs autograd or already lives on the device takes the plain path """
    from pydens_amd import Solver
    host = type('H', (), {'device': torch.device('cpu'), 'reshape_and_concat': Solver.reshape_and_concat,
                          '_points_on_device': Solver._points_on_device})()
    a = host._points_on_device((torch.tensor([0.5]),))
    b = host._points_on_device((torch.tensor([0.5]),))
    c = host._points_on_device((torch.tensor([0.25]),))
    assert a is b and c is not a and torch.equal(a, torch.tensor([[0.5]])) and torch.equal(c, torch.tensor([[0.25]]))
    assert host._points_on_device((np.array([0.5, 0.75]), 2.0)) is host._points_on_device((np.array([0.5, 0.75]), 2.0))
    assert torch.equal(host._points_on_device((np.array([0.5, 0.75]), 2.0)), torch.tensor([[0.5, 2.0], [0.75, 2.0]]))
    x = torch.tensor([0.5], requires_grad=True)
    assert host._points_on_device((x,)) is not host._points_on_device((x,))        # differentiable inputs are never cached
    big = torch.zeros(5000)
    assert host._points_on_device((big,)) is not host._points_on_device((big,))    # nor are batches of points
    for i in range(80):                                                               # the cache stays bounded
        host._points_on_device((torch.tensor([float(i)]),))
    assert len(host._host_constants) <= 64


def test_autograd_fallbacks_of_D_are_counted():
    """ tokens.AUTOGRAD_FALLBACKS: D on something that is not a kernel stream differentiates by torch autograd (create_graph); the solver
    never records a step in which that happened as a launch graph """
    from pydens_amd import tokens
    x = torch.linspace(0, 1, 5).reshape(-1, 1).requires_grad_()
    before = tokens.AUTOGRAD_FALLBACKS[0]
    d = tokens.D(torch.sin(x), x)
    assert tokens.AUTOGRAD_FALLBACKS[0] == before + 1
    assert torch.allclose(d, torch.cos(x))
