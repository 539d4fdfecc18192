""" The experimental chain kernel (tools/experiments/pinn_chain_kernel.h, build knob -DPINN_CHAIN=1: wave-private points, layers
chained through the MFMA accumulators, a weight-gradient wave beside every chain wave, LDS flags instead of barriers) on the
emulated kernels against the oracle: BASELINE configs 2 and 4 (the two shapes it is built for), ragged batches, several
workgroup rounds. The product keeps pinn_tile_kernel for these shapes (measured slower on MI355X); the test
keeps the experiment honest. """
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
import pinn_configs as pc
from conftest import params_close
from helpers import FixedBatches, export_params, load_params


# an experiment outside the product (tools/experiments/): not part of the default CPU suite (a second emulator build + 35 s of emulation)
pytestmark = pytest.mark.skipif(os.environ.get('PINN_TEST_EXPERIMENTS', '0') != '1', reason='experiment kernels: run with PINN_TEST_EXPERIMENTS=1')


@pytest.fixture(scope='module')
def chain_lib():
    import build_emu
    from pydens_amd import engine
    path = build_emu.build(extra_flags=['-DPINN_CHAIN=1'], tag='chain')     # (only the width-64 instantiations differ)
    return engine.bind(ctypes.CDLL(path))


@pytest.mark.parametrize('name,n', [('cfg4', 333), ('cfg2', 16 * 4 * 3 + 5)])
def test_chain_kernel_matches_the_oracle(chain_lib, name, n):
    import pydens_amd as pa
    from oracle import pinn_oracle as po
    torch.manual_seed(0)
    co, cp = pc.make_config(name, po.D, torch), pc.make_config(name, pa.D, torch)
    oracle = po.OracleSolver(co['equation'], **co['solver_kwargs'])
    solver = pa.Solver(cp['equation'], **cp['solver_kwargs'], _lib=chain_lib, device='cpu')
    load_params(solver, oracle.export_params())
    pts = pc.sample_points(co, n, seed=3, steps=3)
    oracle.fit(niters=3, batch_size=n, points=pts, lr=0.01)
    solver.fit(niters=3, batch_size=n, sampler=FixedBatches(pts), lr=0.01)
    assert solver.last_fit_path == 'fused'
    assert chain_lib.pinn_last_kernel_name().decode().startswith('pinn_chain_kernel<')
    np.testing.assert_allclose([float(v) for v in solver.losses], [float(v) for v in oracle.losses], rtol=2e-6)
    for got, want in zip(export_params(solver), oracle.export_params()):
        assert params_close(got, want, 2e-6)


def test_experiment_build_knobs_keep_parity():
    """ the measured-and-rejected build options that are still in the tree (team-local LDS arrival counters
    instead of s_barrier in the two-team kernels; section 6a: no prefetch of the saved jets) change the order of nothing that is summed: the step equals the default build's (and the oracle's loss) """
    import build_emu
    import pydens_amd as pa
    from pydens_amd import engine
    from oracle import pinn_oracle as po
    knobs = build_emu.build(extra_flags=['-DPINN_TEAM_FLAGS=3', '-DPINN_SVPF_MAX=0'], tag='knobs', widths=(64,))
    libs = [engine.bind(ctypes.CDLL(build_emu.build())), engine.bind(ctypes.CDLL(knobs))]
    for name, n in (('cfg2', 70), ('cfg4', 70)):              # the two-team kernels (team-local barriers), 64-wide
        torch.manual_seed(0)
        co, cp = pc.make_config(name, po.D, torch), pc.make_config(name, pa.D, torch)
        oracle = po.OracleSolver(co['equation'], **co['solver_kwargs'])
        pts = pc.sample_points(co, n, seed=5, steps=2)
        runs = []
        for lib in libs:
            solver = pa.Solver(cp['equation'], **cp['solver_kwargs'], _lib=lib, device='cpu')
            load_params(solver, oracle.export_params())
            solver.fit(niters=2, batch_size=n, sampler=FixedBatches(pts), lr=0.01)
            assert solver.last_fit_path == 'fused'
            runs.append(([float(v) for v in solver.losses], export_params(solver)))
        oracle.fit(niters=2, batch_size=n, points=pts, lr=0.01)
        np.testing.assert_allclose(runs[1][0], [float(v) for v in oracle.losses], rtol=3e-6)
        np.testing.assert_allclose(runs[1][0], runs[0][0], rtol=1e-6)
        for got, want in zip(runs[1][1], runs[0][1]):
            assert params_close(got, want, 1e-6)
