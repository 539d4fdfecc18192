""" The split-bf16 GEMM variant (pinn_tile_kernel VAR 512; include/pinn.h pinn_set_gemm_mode) on the CPU emulator: same kernel
source as libpinn_hip.so with an emulated v_mfma_f32_16x16x32_bf16 / ds_read_b64_tr_b16 (tests/emu). Every fp32 operand of the
hidden-layer GEMMs is split exactly into three bf16 and six partial products are accumulated in fp32, so the variant is held to
the SAME fixtures and tolerances as the exact-fp32 kernels (reference-generated goldens; model_torch.py:170-178, :460). """
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

import pinn_configs as pc
from conftest import Golden, params_close, rel_l2
from helpers import FixedBatches, export_grads, export_params, fit_rtol, load_params, make_solver, ran_split_kernel

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))


@pytest.fixture(scope='module')
def emu_lib():
    import build_emu
    from pydens_amd import engine
    return engine.bind(ctypes.CDLL(build_emu.build()))


@pytest.fixture(scope='module')
def pa():
    import pydens_amd
    return pydens_amd


@pytest.mark.parametrize('name', ['cfg2', 'cfg4', 'cfg3', 'cfg5'])
def test_split_kernels_follow_the_reference_goldens(pa, emu_lib, name):
    g = Golden(name)
    _, solver = make_solver(name, pa, gemm='bf16x3', _lib=emu_lib, device='cpu')
    load_params(solver, g.params)
    xs = torch.from_numpy(g.points[0].copy())
    solver._fused_step(xs, 1)
    assert ran_split_kernel(solver)
    if name in ('cfg3', 'cfg5'):                    # widths >= 128: the streamed weight-gradient kernel in its split form as well
        assert emu_lib.pinn_last_wgrad_kernel_name().decode().rstrip('>').split(',')[5] == 'true'      # <HP,ND,N2,COMB,MT,SPLIT,SKIPS,HEAVY>
    lay = solver.model.net.layout
    assert abs(float(solver.grads[lay.off_loss]) - g.loss0) <= 1e-5 * g.loss0
    for got, want in zip(export_grads(solver), g.grads):
        if want is None:
            assert float(np.abs(got).max()) == 0.0
        else:
            assert rel_l2(got, want) < 1e-5
    niters = {'cfg3': 1, 'cfg5': 1}.get(name, len(g.losses))           # 8 waves x 128 / 256 units are slow to emulate: one step suffices
    solver.fit(niters=niters, batch_size=g.points.shape[1], sampler=FixedBatches(g.points), lr=g.lr)
    assert solver.last_fit_path == 'fused' and ran_split_kernel(solver)
    np.testing.assert_allclose([float(v) for v in solver.losses], g.losses[:niters], rtol=fit_rtol(name))
    if niters == len(g.losses):
        for got, want in zip(export_params(solver), g.finals):
            assert params_close(got, want, fit_rtol(name))


@pytest.mark.parametrize('name', ['cfg2', 'cfg4'])
@pytest.mark.parametrize('n', [1, 15, 17, 33, 100])
def test_split_kernels_on_ragged_batches(pa, emu_lib, name, n):
    """ tail tiles, a lone team, empty rounds of the second team: the split kernel against the exact one on the same points """
    torch.manual_seed(3)
    cfg, solver = make_solver(name, pa, _lib=emu_lib, device='cpu')
    pts = torch.from_numpy(pc.sample_points(cfg, n, seed=21))
    out = {}
    for gemm in ('fp32', 'bf16x3'):
        solver.set_gemm_mode(gemm)
        solver._fused_step(pts, 1)
        assert ran_split_kernel(solver) == (gemm == 'bf16x3')
        out[gemm] = solver.grads.clone().numpy()
    lay = solver.model.net.layout
    assert abs(out['bf16x3'][lay.off_loss] - out['fp32'][lay.off_loss]) <= 2e-6 * abs(out['fp32'][lay.off_loss])
    assert rel_l2(out['bf16x3'][:lay.p_core], out['fp32'][:lay.p_core]) < 2e-6


def test_shapes_without_a_split_kernel_keep_the_fp32_kernels(pa, emu_lib):
    """ the mode is a request: nets / problems the split kernels are not built for run the exact kernels, unchanged """
    for name in ('cfg1', 'ode_sigmoid'):
        g = Golden(name)
        _, solver = make_solver(name, pa, _lib=emu_lib, device='cpu')
        load_params(solver, g.params)
        xs = torch.from_numpy(g.points[0].copy())
        solver._fused_step(xs, 1)
        want = solver.grads.clone()
        solver.set_gemm_mode('bf16x3')
        solver._fused_step(xs, 1)
        assert not ran_split_kernel(solver)
        assert torch.equal(solver.grads, want)
    with pytest.raises(ValueError):
        solver.set_gemm_mode('fp8')
    assert emu_lib.pinn_set_gemm_mode(solver.model.net.handle, 7) != 0
    assert b'unknown GEMM mode' in emu_lib.pinn_last_error()


def test_split_mode_survives_the_generic_path_and_predict(pa, emu_lib):
    """ forward-only and backward-only launches (predict, the generic step path) have no split instantiation: same results """
    g = Golden('cfg2')
    _, solver = make_solver('cfg2', pa, gemm='bf16x3', _lib=emu_lib, device='cpu')
    load_params(solver, g.params)
    pts = g.points
    pred = solver.predict(*[pts[1][:, i] for i in range(pts.shape[2])])
    assert np.abs(pred[:, 0] - g.predict).max() <= 1e-5 * max(1.0, np.abs(g.predict).max())
    solver.program = None
    solver.fit(niters=2, batch_size=pts.shape[1], sampler=FixedBatches(pts[:2]), lr=g.lr)
    assert solver.last_fit_path == 'generic'
    np.testing.assert_allclose([float(v) for v in solver.losses], g.losses[:2], rtol=fit_rtol('cfg2'))
