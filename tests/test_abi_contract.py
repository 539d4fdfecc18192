""" Error behaviour of the C-ABI (include/pinn.h): every entry point returns non-zero and leaves a message in
pinn_last_error() on bad input -- nothing throws, nothing is launched, nothing is written. Exercised through the same
pinn_abi.cpp compiled for the emulator (no GPU needed); the reference's own error surface (ValueError on a malformed
domain, model_torch.py:43-45) is covered in test_host.py. """
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))


@pytest.fixture(scope='module')
def lib():
    import build_emu
    from pydens_amd import engine
    return engine.bind(ctypes.CDLL(build_emu.build()))


def _err(lib):
    return lib.pinn_last_error().decode()


def _net(lib, dims=(2, 16, 16, 1), **kw):
    from pydens_amd import engine
    return engine.Net(list(dims), 'tanh', kw.pop('ndims', 2), lib=lib, **kw)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def test_create_rejects_what_the_kernels_do_not_instantiate(lib):
    from pydens_amd import engine
    for dims, kw, needle in (((2, 600, 1), {}, 'width'),                      # wider than 512
                             ((2,) + (300,) * 9 + (1,), {}, 'at most 8 hidden layers'),     # width 512: eight bias-gradient rows in the LDS carve (round 6)
                             ((2,) + (8,) * 33 + (1,), {}, 'layer'),           # more than PINN_MAX_LAYERS (32) linear layers
                             ((9, 8, 1), dict(ndims=9), 'ndims+nparams'),     # more than PINN_MAX_INPUTS columns
                             ((2, 8, 2), {}, 'one unit')):                     # not a scalar field
        with pytest.raises(RuntimeError) as err:
            _net(lib, dims, **kw)
        assert needle in str(err.value).lower(), str(err.value)
    with pytest.raises(NotImplementedError):
        engine.Net([2, 8, 1], 'hardswish', 2, lib=lib)           # (an activation outside the sixteen codes of include/pinn.h)
    # round 5: ReLU & co. exist -- ReLU (code 7) with the first set of full breadth kernels, codes above 7 on the second set
    assert not engine.Net([2, 8, 1], 'relu', 2, lib=lib).allact and engine.Net([2, 8, 1], 'elu', 2, lib=lib).allact
    # round 6: widths 257 .. 512 exist (padded to 512); what that width does not carry is refused where the streams are planned
    wide = _net(lib, (2, 300, 300, 1))
    assert wide.layout.hp == 512
    from pydens_amd import trace
    spec = trace.StreamSpec({(0,), (0, 0), (1,), (1, 1), (2,)}, hp=512)
    assert [g[1] for g in spec.groups] == [1, 1, 0] and [len(g[0]) for g in spec.groups] == [1, 1, 1] and not spec.single_call and not spec.combinable
    assert trace.StreamSpec({(0,), (0, 0)}, hp=512).single_call and trace.StreamSpec({(0,), (1,)}, hp=512).single_call
    with pytest.raises(NotImplementedError, match='width 512'):
        trace.StreamSpec({(0,), (0, 0), (0, 0, 0)}, hp=512)
    with pytest.raises(NotImplementedError, match='second set'):
        trace.StreamSpec({(0,)}, hp=512, allact=True)
    bad = (ctypes.c_int * 3)(2, 8, 1)
    assert lib.pinn_create_ex(bad, 2, None, 0, None, None, 2, 0, 0, 0, None, None, 0.0, None) != 0


def test_activation_parameters_are_validated(lib):
    """ pinn_set_act_params (round 6): one value per activation, only where the activation takes one, Softplus beta positive """
    from pydens_amd import engine
    net = engine.Net([2, 8, 8, 1], ['LeakyReLU:0.2', 'Softplus:2.0'], 2, lib=lib)
    assert net.act_params == [0.2, 2.0] and net.allact
    f3 = ctypes.c_float * 3
    assert lib.pinn_set_act_params(net.handle, f3(0.2, 2.0, 0.0), 3) != 0 and 'activations' in _err(lib)          # wrong count
    f2 = ctypes.c_float * 2
    assert lib.pinn_set_act_params(net.handle, f2(0.2, -1.0), 2) != 0 and 'beta' in _err(lib)
    assert lib.pinn_set_act_params(net.handle, f2(0.3, 1.5), 2) == 0
    tanh = engine.Net([2, 8, 8, 1], 'tanh', 2, lib=lib)
    assert lib.pinn_set_act_params(tanh.handle, f2(0.5, 0.0), 2) != 0 and 'takes no parameter' in _err(lib)
    assert lib.pinn_set_act_params(tanh.handle, f2(0.0, 0.0), 2) == 0
    with pytest.raises(NotImplementedError):
        engine.Net([2, 8, 1], 'Tanh:0.5', 2, lib=lib)


def test_layout_is_consistent(lib):
    net = _net(lib, (3, 20, 20, 20, 1), ndims=2, nparams=1)
    lay = net.layout
    assert lay.hp == 32 and lay.lh == 2 and lay.d == 3
    assert lay.off_w1 == 0 and lay.off_b1 == lay.hp * lay.d and lay.off_wh == lay.off_b1 + lay.hp
    assert lay.hidden_stride == lay.hp * lay.hp + lay.hp and lay.off_wl == lay.off_wh + lay.lh * lay.hidden_stride
    assert lay.off_bl == lay.off_wl + lay.hp and lay.off_log_scale == lay.off_bl + 1 and lay.off_loss == lay.off_bl + 2
    assert lay.p_core % 4 == 0 and lay.off_extra == lay.p_core and lay.p_total == lay.p_core + 16
    a, b = net.workspace_bytes(1000, 2, 2), net.workspace_bytes(100000, 2, 2)
    assert 0 < a <= b and net.workspace_bytes(1000, 5, 0) == 0                # nd > PINN_MAX_DIRS: no plan


def test_forward_and_backward_argument_checks(lib):
    net = _net(lib)
    lay = net.layout
    params = torch.zeros(lay.p_total)
    xs = torch.rand(10, 2)
    out = torch.zeros(1, 10)
    dirs = (ctypes.c_int * 1)(0)
    assert lib.pinn_jet_forward(net.handle, None, _ptr(xs), 10, dirs, 0, 0, None, 0.0, _ptr(out), None) != 0
    assert 'null' in _err(lib)
    assert lib.pinn_jet_forward(net.handle, _ptr(params), _ptr(xs), 10, dirs, 5, 0, None, 0.0, _ptr(out), None) != 0
    bad_dir = (ctypes.c_int * 1)(7)                                          # column 7 of a 2-column problem
    assert lib.pinn_jet_forward(net.handle, _ptr(params), _ptr(xs), 10, bad_dir, 1, 0, None, 0.0, _ptr(out), None) != 0
    assert lib.pinn_jet_forward(net.handle, _ptr(params), _ptr(xs), 0, dirs, 0, 0, None, 0.0, _ptr(out), None) == 0   # empty batch
    grads = torch.zeros(lay.p_total)
    gin = torch.zeros(1, 10)
    ws = torch.zeros(net.workspace_bytes(10, 0, 0) // 4 + 8)
    assert lib.pinn_jet_backward(net.handle, _ptr(params), _ptr(xs), 10, dirs, 0, 0, None, 0.0, _ptr(gin), _ptr(grads), 0,
                                 _ptr(ws), 16, None) != 0
    assert 'workspace too small' in _err(lib)
    assert lib.pinn_jet_backward(net.handle, _ptr(params), _ptr(xs), 10, dirs, 0, 0, None, 0.0, _ptr(gin), _ptr(grads), 0,
                                 ctypes.c_void_p(ws.data_ptr() + 4), ws.numel() * 4 - 4, None) != 0
    assert 'aligned' in _err(lib)
    assert torch.count_nonzero(grads) == 0                                    # nothing ran


def test_residual_step_validates_the_program(lib):
    from pydens_amd import engine
    from pydens_amd.engine import OPS, Residual
    net = _net(lib)
    lay = net.layout
    params, grads = torch.zeros(lay.p_total), torch.zeros(lay.p_total)
    xs = torch.rand(10, 2)
    ws = torch.zeros(net.workspace_bytes(10, 0, 0) // 4 + 8)
    dirs = (ctypes.c_int * 1)(0)

    def step(res, n=10, fn=lib.pinn_residual_step):
        return fn(net.handle, ctypes.byref(res), _ptr(params), _ptr(xs), n, dirs, 0, 0, None, 0.0, 0.1, _ptr(grads),
                  _ptr(ws), ws.numel() * 4, None)
    first = 1 + 2                                                             # S + d: first temporary register
    good = Residual.build(engine.RES_PROGRAM, 0, None, program=([(OPS['MUL'], first, 0, 0)], []))
    assert step(good) == 0 and float(grads[lay.off_loss]) >= 0.0
    assert step(good, fn=lib.pinn_residual_step_add) == 0
    assert step(good, n=0) != 0 and 'n_points' in _err(lib)
    cases = [
        (Residual.build(engine.RES_PROGRAM, 0, None, program=([], [])), 'empty'),
        (Residual.build(engine.RES_PROGRAM, 0, None, program=([(99, first, 0, 0)], [])), 'opcode'),
        (Residual.build(engine.RES_PROGRAM, 0, None, program=([(OPS['MUL'], 0, 0, 0)], [])), 'overwrites'),
        (Residual.build(engine.RES_PROGRAM, 0, None, program=([(OPS['MUL'], first, 200, 0)], [])), 'register'),
        (Residual.build(engine.RES_PROGRAM, 0, None, program=([(OPS['CONST'], first, 3, 0)], [1.0])), 'constant'),
        (Residual.build(engine.RES_PROGRAM, 0, None, program=([(OPS['STORE'], 0, 0, 0)], [])), 'store'),
        (Residual.build(engine.RES_AFFINE, 0, None, coef=[1.0], coef_row=[2]), 'pre-pass row'),
        (Residual.build(engine.RES_AFFINE, 0, None, coef=[1.0], coef_row=[-1], src_row=0), 'pre-pass row'),
    ]
    for res, needle in cases:
        assert step(res) != 0, needle
        assert needle in _err(lib).lower(), (needle, _err(lib))
    bad = Residual.build(engine.RES_PROGRAM, 0, None, program=([(OPS['MUL'], first, 0, 0)], []))
    bad.kind = 7
    assert step(bad) != 0 and 'kind' in _err(lib)
    bad = Residual.build(engine.RES_PROGRAM, 0, None, program=([(OPS['MUL'], first + 9, 0, 0)], []), n_vars=9)
    assert step(bad) != 0 and 'n_vars' in _err(lib)
    bad = Residual.build(engine.RES_AFFINE, 0, None, coef=[1.0], coef_row=[-1], n_vars=1)
    assert step(bad) != 0 and 'program' in _err(lib)
    bad = Residual.build(engine.RES_PROGRAM, 0, None, program=([(OPS['MUL'], first, 0, 0)], []))
    bad.ic_var1 = 1                                                           # the problem has no initial condition
    assert step(bad) != 0 and 'ic_var1' in _err(lib)
    bad.ic_var1 = 0
    bad.n_aux = 40
    assert step(bad) != 0 and 'n_aux' in _err(lib)


def test_sampler_and_adam_argument_checks(lib):
    xs = torch.zeros(8, 3)
    kind = (ctypes.c_int * 3)(0, 1, 2)
    a = (ctypes.c_float * 3)(0, 0, 1)
    b = (ctypes.c_float * 3)(1, 1, 0)
    assert lib.pinn_sample_points(_ptr(xs), 8, 3, kind, a, b, 1, 0, None) == 0
    assert lib.pinn_sample_points(_ptr(xs), 0, 3, kind, a, b, 1, 0, None) == 0
    assert lib.pinn_sample_points(None, 8, 3, kind, a, b, 1, 0, None) != 0
    assert lib.pinn_sample_points(_ptr(xs), 8, 9, kind, a, b, 1, 0, None) != 0 and 'd=' in _err(lib)
    bad_kind = (ctypes.c_int * 3)(0, 5, 2)
    assert lib.pinn_sample_points(_ptr(xs), 8, 3, bad_kind, a, b, 1, 0, None) != 0 and 'kind' in _err(lib)
    p, g, m, v = (torch.zeros(16) for _ in range(4))
    step = torch.zeros(1, dtype=torch.int32)
    assert lib.pinn_adam_step(_ptr(p), None, _ptr(m), _ptr(v), None, 16, _ptr(step), 0.1, 0.9, 0.999, 1e-8, None) != 0
    assert lib.pinn_adam_step_at(_ptr(p), _ptr(g), _ptr(m), _ptr(v), None, 16, _ptr(step), 0, 0.1, 0.9, 0.999, 1e-8, None, 0, None) != 0
    assert lib.pinn_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), None, 0, _ptr(step), 0.1, 0.9, 0.999, 1e-8, None) == 0
    assert np.all(p.numpy() == 0)
