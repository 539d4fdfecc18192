""" GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C-ABI by the Python
host, against (a) the golden fixtures the UNMODIFIED reference produced and (b) the oracle restatement run on the
box's CPU, on identical parameters and points. Tolerances: 1e-5 relative on loss and predicted field
(BASELINE.json north_star), fp32; gradients 1e-5 relative L2 per tensor (helpers.GRAD_RTOL).
The tests of the two BASELINE width-64 shapes run twice: exact-fp32 GEMMs and the gated split-bf16 variant (`gemm`). """
import ctypes
import os

import numpy as np
import pytest
import torch

import pinn_configs as pc
from conftest import Golden, params_close, rel_l2
from helpers import (GRAD_RTOL, FixedBatches, close_or_arbitrated, export_grads, export_params, fit_rtol, grad_close, load_params,
                     make_solver, ran_split_kernel, record_margin)

pytestmark = pytest.mark.gpu

SUPPORTED = ('cfg1', 'cfg2', 'cfg3', 'cfg4', 'cfg5', 'ode_sigmoid', 'mixed', 'heat3d', 'kdv', 'resnet3')


@pytest.fixture(scope='module')
def pa():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    import pydens_amd
    from pydens_amd import engine
    assert engine.load_library().pinn_backend() == b'hip-gfx950'
    return pydens_amd


SPLIT_SHAPES = ('cfg2', 'cfg3', 'cfg4', 'cfg5')          # the shapes the split-bf16 kernels are built for
WITH_GEMM = [(n, 'fp32') for n in SUPPORTED] + [(n, 'bf16x3') for n in SPLIT_SHAPES]


@pytest.mark.parametrize('name,gemm', WITH_GEMM)
def test_predict_and_step_match_reference_golden(pa, name, gemm):
    g = Golden(name)
    _, solver = make_solver(name, pa, gemm=gemm)
    load_params(solver, g.params)
    pts = g.points
    pred = solver.predict(*[pts[1][:, i] for i in range(pts.shape[2])])
    assert pred.shape == (pts.shape[1], 1) and pred.dtype == np.float32
    assert np.abs(pred[:, 0] - g.predict).max() <= 1e-5 * max(1.0, np.abs(g.predict).max())

    # one fused residual+grad evaluation on batch 0 (no optimizer step): loss and every parameter gradient
    assert solver.program is not None, solver.program_error
    xs = torch.from_numpy(pts[0].copy()).cuda()
    solver._fused_step(xs, 1)
    assert ran_split_kernel(solver) == (gemm == 'bf16x3')
    lay = solver.model.net.layout
    loss = float(solver.grads[lay.off_loss])
    assert abs(loss - g.loss0) <= 1e-5 * g.loss0
    for got, want in zip(export_grads(solver), g.grads):
        if want is None:
            assert float(np.abs(got).max()) == 0.0
        else:
            assert grad_close(got, want)


@pytest.mark.parametrize('name,gemm', WITH_GEMM)
@pytest.mark.parametrize('path', ['fused', 'generic'])
def test_fit_matches_reference_golden(pa, name, gemm, path):
    if gemm == 'bf16x3' and path == 'generic':
        pytest.skip('the generic path has no split kernels (forward / backward launches)')
    g = Golden(name)
    _, solver = make_solver(name, pa, gemm=gemm)
    load_params(solver, g.params)
    if path == 'generic':
        solver.program = None
    solver.fit(niters=len(g.losses), batch_size=g.points.shape[1], sampler=FixedBatches(g.points), lr=g.lr)
    assert solver.last_fit_path == path
    losses = np.array([float(v) for v in solver.losses])
    np.testing.assert_allclose(losses, g.losses, rtol=fit_rtol(name))
    for got, want in zip(export_params(solver), g.finals):
        assert rel_l2(got, want) < fit_rtol(name)


def test_streams_match_fp64_jets(pa):
    """ the derivative streams D(...) resolves to, against the fp64 jet formulas, heat config (IC + BC, 6 streams) """
    from oracle import jet_f64 as jf, problems
    from test_oracle_vs_golden import make_spec
    g = Golden('cfg3')
    _, solver = make_solver('cfg3', pa)
    load_params(solver, g.params)
    sf, spec = make_spec(g)
    pts = g.points[0]
    out = jf.step(spec, pts, sf['residual'], problems.ic_streams_f64('cfg3', pts, sf['dir_cols'], sf['n2']))
    xs = torch.from_numpy(pts.copy()).cuda()
    ic = torch.from_numpy(problems.ic_streams_f64('cfg3', pts, sf['dir_cols'], sf['n2']).astype(np.float32)).cuda()
    streams = solver.model.net.jet_forward(solver.model.flat, xs, sf['dir_cols'], sf['n2'], ic_streams=ic.contiguous())
    for s in range(spec.S):
        assert rel_l2(streams[s].cpu().numpy(), out['u_streams'][s]) < 2e-5, s


@pytest.mark.parametrize('gemm', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('n', [1, 15, 17, 1000, 4099])
def test_ragged_batch_sizes(pa, n, gemm):
    """ tiles of 16 points: every tail length must give the same per-point answer and a correctly normalised loss """
    g = Golden('cfg2')
    cfg, solver = make_solver('cfg2', pa, gemm=gemm)
    load_params(solver, g.params)
    pts = pc.sample_points(cfg, 4112, seed=11)
    full = solver.predict(pts[:, 0], pts[:, 1])
    part = solver.predict(pts[:n, 0], pts[:n, 1])
    assert np.array_equal(full[:n], part)
    lay = solver.model.net.layout
    xs = torch.from_numpy(pts[:n].copy()).cuda()
    solver._fused_step(xs, 1)
    loss_fused = float(solver.grads[lay.off_loss])
    streams = solver.model.net.jet_forward(solver.model.flat, xs, [0, 1], 2)
    r = streams[3] + streams[4] - 5 * torch.sin(np.pi * (xs[:, 0] + xs[:, 1]))
    assert abs(loss_fused - float((r * r).mean())) <= 2e-6 * loss_fused


@pytest.mark.parametrize('gemm', ['fp32', 'bf16x3'])
def test_full_size_step_against_chunked_oracle(pa, gemm):
    """ BASELINE config 2 at its full batch (65 536 points): loss and gradients vs the oracle evaluated in chunks """
    from oracle import pinn_oracle as po
    torch.manual_seed(5)
    cfg, solver = make_solver('cfg2', pa, gemm=gemm)
    ocfg = pc.make_config('cfg2', po.D, torch)
    oracle = po.OracleSolver(ocfg['equation'], **ocfg['solver_kwargs'])
    oracle.import_params(export_params(solver))
    pts = pc.sample_points(cfg, cfg['n_points'], seed=2)
    ev = oracle.evaluate(pts, chunk=16384)
    solver._fused_step(torch.from_numpy(pts).cuda(), 1)
    lay = solver.model.net.layout
    assert abs(float(solver.grads[lay.off_loss]) - ev['loss']) <= 1e-5 * ev['loss']
    for got, want in zip(export_grads(solver), oracle.export_grads()):
        if want is not None:
            assert grad_close(got, want)


@pytest.mark.parametrize('gemm', ['fp32', 'bf16x3'])
def test_sharded_sum_equals_whole(pa, gemm):
    """ data-parallel property: gradients of shards, each scaled by 1/N_global, add up to the whole """
    g = Golden('cfg4')
    cfg, solver = make_solver('cfg4', pa, gemm=gemm)
    load_params(solver, g.params)
    pts = torch.from_numpy(pc.sample_points(cfg, 4096, seed=4)).cuda()
    solver._fused_step(pts, 1)
    whole = solver.grads.clone()
    acc = torch.zeros_like(whole)
    for shard in pts.chunk(4):
        solver._fused_step(shard.contiguous(), 4)
        acc += solver.grads
    assert rel_l2(acc.cpu().numpy(), whole.cpu().numpy()) < 1e-5


@pytest.mark.parametrize('name,n_full,shards', [('cfg3', 262144, 4), ('cfg4', 1048576, 8), ('cfg5', 1048576, 8)])
def test_baseline_full_sizes_through_size_independent_properties(pa, name, n_full, shards):
    """ BASELINE configs 3-5 at the batch BASELINE.json quotes (262 144 / 1 048 576 points: too large for the oracle in a test
    run): (a) the gradients + loss of `shards` equal shards, each scaled by 1/N_global, add up to the whole batch -- exactly
    what the RCCL all-reduce of the 8-GPU configs computes; (b) shuffling the points changes nothing beyond summation order;
    (c) a 4 096-point prefix agrees with the oracle, which ties the large run to the reference arithmetic. """
    from oracle import pinn_oracle as po
    torch.manual_seed(9)
    cfg, solver = make_solver(name, pa)
    pts = torch.from_numpy(pc.sample_points(cfg, n_full, seed=6)).cuda()
    lay = solver.model.net.layout
    solver._fused_step(pts, 1)
    whole = solver.grads.clone()
    assert torch.isfinite(whole).all() and float(whole[lay.off_loss]) > 0
    acc = torch.zeros_like(whole)
    for shard in pts.chunk(shards):
        solver._fused_step(shard.contiguous(), shards)
        acc += solver.grads
    assert rel_l2(acc[:lay.p_core].cpu().numpy(), whole[:lay.p_core].cpu().numpy()) < 1e-5
    assert abs(float(acc[lay.off_loss]) - float(whole[lay.off_loss])) <= 1e-5 * float(whole[lay.off_loss])
    perm = torch.randperm(n_full, device=pts.device)
    solver._fused_step(pts[perm].contiguous(), 1)
    assert rel_l2(solver.grads[:lay.p_core].cpu().numpy(), whole[:lay.p_core].cpu().numpy()) < 1e-5
    ocfg = pc.make_config(name, po.D, torch)
    oracle = po.OracleSolver(ocfg['equation'], **ocfg['solver_kwargs'])
    oracle.import_params(export_params(solver))
    head = pts[:4096].contiguous()
    ev = oracle.evaluate(head.cpu().numpy(), chunk=2048)
    solver._fused_step(head, 1)
    assert abs(float(solver.grads[lay.off_loss]) - ev['loss']) <= 1e-5 * ev['loss']
    for got, want in zip(export_grads(solver), oracle.export_grads()):
        if want is not None:
            assert grad_close(got, want)


@pytest.mark.parametrize('name,iters,batch,gemm', [('cfg2', 600, 4096, 'fp32'), ('cfg4', 400, 4096, 'fp32'), ('cfg3', 150, 2048, 'fp32'),
                                                   ('cfg2', 600, 4096, 'bf16x3'), ('cfg4', 400, 4096, 'bf16x3'), ('cfg3', 150, 2048, 'bf16x3')])
def test_trained_models_against_the_fp64_arbiter(pa, name, iters, batch, gemm):
    """ on a TRAINED model the residual is a small difference of large terms and the reference's own
    fp32 result is only good to 1e-5 .. 1e-2 (gradients of cfg3!) of the fp64 value, so neither engine can be held to 1e-5
    of the other; the fp64 oracle arbitrates: |ours - f64| <= max(k |ref32 - f64|, 1e-5 |f64|) with k = 2 for the loss and
    the predicted field AND for the gradient (all tensors as one vector; the survey's rule, SURVEY 8c item 5 -- round 1 ran
    the gradient at k = 3 because its tanh lost relative accuracy for small arguments) after `iters`
    Adam iterations of Solver.fit on the device. """
    from oracle import pinn_oracle as po
    torch.manual_seed(13)
    cfg, solver = make_solver(name, pa, gemm=gemm)
    sampler = pa.NumpySampler('uniform') & pa.NumpySampler('uniform', low=1, high=5) if name == 'cfg4' else None
    solver.fit(niters=iters, batch_size=batch, sampler=sampler, lr=0.005)
    losses = solver.losses
    assert float(losses[-1]) < 0.25 * float(losses[0])              # it did train
    params = export_params(solver)
    ocfg = pc.make_config(name, po.D, torch)
    evals = {}
    pts = pc.sample_points(cfg, 4096, seed=17)
    for dtype in (torch.float32, torch.float64):
        oracle = po.OracleSolver(ocfg['equation'], **ocfg['solver_kwargs'], dtype=dtype)
        oracle.import_params(params)
        ev = oracle.evaluate(pts, chunk=2048)
        evals[dtype] = (ev['loss'], oracle.export_grads(), ev['u'])
    solver._fused_step(torch.from_numpy(pts).cuda(), 1)
    lay = solver.model.net.layout
    (l32, g32, u32), (l64, g64, u64) = evals[torch.float32], evals[torch.float64]

    def flat(tensors, mask):
        return np.concatenate([np.asarray(t, dtype=np.float64).ravel() for t, m in zip(tensors, mask) if m is not None])

    def within(ours, ref32, ref64, k, floor=1e-5):
        ours, ref32, ref64 = (np.asarray(v, dtype=np.float64).ravel() for v in (ours, ref32, ref64))
        err, ref_err, scale = (float(np.linalg.norm(v)) for v in (ours - ref64, ref32 - ref64, ref64))
        return err <= max(k * ref_err, floor * scale), (err, ref_err, scale)
    # (loss of a trained model = a mean of 4096 squared residuals of size 1e-2: measured 7e-6 / 2e-6 / 5e-8 relative on
    #  cfg2 / cfg4 / cfg3; the floor leaves the margin a different summation order may need)
    ok, detail = within(float(solver.grads[lay.off_loss]), l32, l64, 2.0, floor=2e-5)
    assert ok, ('loss', detail)
    ok, detail = within(flat(export_grads(solver), g64), flat(g32, g64), flat(g64, g64), 2.0)
    assert ok, ('gradient', detail)
    u = solver.predict(*[pts[:, c] for c in range(pts.shape[1])])
    ok, detail = within(u, u32, u64, 2.0)
    assert ok, ('field', detail)


def test_adam_matches_torch(pa):
    from pydens_amd import engine
    torch.manual_seed(0)
    n = 5000
    p = torch.randn(n, device='cuda'); ref = p.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=0.01)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    step = torch.zeros(1, dtype=torch.int32, device='cuda')
    mask = torch.ones(n, dtype=torch.uint8, device='cuda'); mask[::7] = 0
    keep = p.clone()
    net = engine.Net([2, 16, 1], 'tanh', 2)
    for k in range(20):
        grad = torch.randn(n, device='cuda')
        ref.grad = grad.clone()
        opt.step()
        # odd steps: the count lives on the device (two launches); even steps: the host passes it (one launch)
        net.adam_step(p, grad, m, v, mask, step, 0.01, at=0 if k % 2 == 0 else k + 1)
    live = mask.bool()
    assert torch.equal(p[~live], keep[~live])
    assert rel_l2(p[live].cpu().numpy(), ref.detach()[live].cpu().numpy()) < 1e-6
    assert int(step) == 20


@pytest.mark.parametrize('which', ['nonlinear', 'variable_coefficient', 'mixed_affine', 'mixed_nonlinear', 'divergence_form',
                                   'conservative_burgers'])
def test_residual_kinds_match_the_oracle(pa, which):
    """ residual PROGRAM (nonlinear Burgers-type, interpreter inside the tile kernel) and AFFINE residual with
    x-dependent coefficients (pre-pass rows) against the oracle's nested autograd on the same points """
    from oracle import pinn_oracle as po
    import test_emu_engine as te
    problem, kind = dict(nonlinear=(te._nonlinear_problem, 0), variable_coefficient=(te._variable_coefficient_problem, 1),
                         mixed_affine=(te._mixed_affine_problem, 1), mixed_nonlinear=(te._mixed_nonlinear_problem, 0),
                         divergence_form=(te._divergence_form_problem, 1),
                         conservative_burgers=(te._conservative_burgers_problem, 0))[which]
    eq_o, kw = problem(po.D, torch)
    oracle = po.OracleSolver(eq_o, **kw)
    eq_p, kw = problem(pa.D, torch)
    solver = pa.Solver(eq_p, **kw)
    assert solver.program is not None and solver.residual_plan.kind == kind, solver.program_error
    start = oracle.export_params()
    load_params(solver, start)
    pts = np.random.RandomState(8).rand(4, 1000, 2).astype(np.float32)
    oracle.fit(niters=4, batch_size=1000, points=pts, lr=0.01)
    solver.fit(niters=4, batch_size=1000, sampler=FixedBatches(pts), lr=0.01)
    assert solver.last_fit_path == 'fused'
    arbiter = {}

    def o64():          # four Adam steps of the reference in fp64 from the same start: the arbiter where its fp32 trajectory is the noisy one
        if not arbiter:
            o = po.OracleSolver(problem(po.D, torch)[0], dtype=torch.float64, **problem(po.D, torch)[1])
            o.import_params(start)
            o.fit(niters=4, batch_size=1000, points=pts, lr=0.01)
            arbiter['o'] = o
        return arbiter['o']
    ok, err, arb = close_or_arbitrated([float(v) for v in solver.losses], [float(v) for v in oracle.losses],
                                       lambda: [float(v) for v in o64().losses], 2e-5, atol=0.0)
    record_margin('residual_kinds', which, 'losses', err, 2e-5, arb)
    assert ok, err
    for i, (got, want) in enumerate(zip(export_params(solver), oracle.export_params())):
        ok, err, arb = close_or_arbitrated(got, want, lambda i=i: o64().export_params()[i], 2e-5, atol=3e-7)
        record_margin('residual_kinds', which, 'parameters', err, 2e-5, arb)
        assert ok, (i, err)


@pytest.mark.parametrize('width', [16, 24, 32, 48, 64, 100, 128, 200, 256])
@pytest.mark.parametrize('depth', [1, 2, 4])
def test_every_width_and_depth_uses_all_its_units(pa, width, depth):
    """ predict, loss and parameter gradients of nets whose hidden layers are exactly `width` wide (all padded lanes
    and K quads of the kernels carry real data) against the oracle: regression test for a width-64 kernel whose last
    K quad was wrong while narrower (zero-padded) nets passed """
    from oracle import pinn_oracle as po
    eq = lambda D: (lambda f, x, t: D(f, t) - 0.1 * D(D(f, x), x) + f * D(f, x))
    kw = dict(ndims=2, boundary_condition=0, initial_condition=lambda x: torch.sin(np.pi * x),
              layout='fa' * depth + 'f', features=[width] * depth + [1], activation='Tanh')
    torch.manual_seed(width * 10 + depth)
    oracle = po.OracleSolver(eq(po.D), **kw)
    solver = pa.Solver(eq(pa.D), **kw)
    assert solver.program is not None, solver.program_error
    load_params(solver, oracle.export_params())
    pts = np.random.RandomState(3).rand(1, 500, 2).astype(np.float32)
    xs = [pts[0][:, i] for i in range(2)]
    assert np.abs(solver.predict(*xs) - oracle.predict(*xs)).max() < 2e-6
    loss_o = oracle.evaluate(pts[0])['loss']
    grads_o = oracle.export_grads()
    arbiter = {}

    def f64():          # the fp64 oracle, evaluated only where the fp32 reference is the noisy side
        if not arbiter:
            o64 = po.OracleSolver(eq(po.D), dtype=torch.float64, **kw)
            o64.import_params(oracle.export_params())
            arbiter['loss'] = o64.evaluate(pts[0])['loss']
            arbiter['grads'] = o64.export_grads()
        return arbiter
    for path in ('fused', 'generic'):
        if path == 'generic':
            solver.program = None
        solver.grads.zero_()
        xs_dev = torch.from_numpy(pts[0].copy()).cuda()
        if path == 'fused':
            solver._fused_step(xs_dev, 1)
        else:
            solver._generic_step(xs_dev, ('equation',), (), torch.nn.MSELoss(), 1)
        lay = solver.model.net.layout
        ok, err, arb = close_or_arbitrated([float(solver.grads[lay.off_loss])], [loss_o], lambda: [f64()['loss']], 1e-5, atol=0.0)
        record_margin('every_width_and_depth', (width, depth, path), 'loss', err, 1e-5, arb)
        assert ok, (path, err)
        for i, (got, want) in enumerate(zip(export_grads(solver), grads_o)):
            if want is not None:
                ok, err, arb = close_or_arbitrated(got, want, lambda i=i: f64()['grads'][i], GRAD_RTOL)
                record_margin('every_width_and_depth', (width, depth, path), 'gradient', err, GRAD_RTOL, arb)
                assert ok, (path, i, err)


@pytest.mark.parametrize('width', [128, 200])
@pytest.mark.parametrize('which', ['poisson', 'burgers', 'poisson_any_activation', 'burgers_any_activation'])
def test_wide_residual_nets_stream_their_weight_gradients(pa, width, which):
    """ skip connections at widths >= 128 (round 4): tile kernel VAR 8 | 1024 | 128 + pinn_wgrad_kernel<..., SKIPS>. A '+' behind an
    activation, one in front of one, a skip from the first layer; 1100 points (69 tiles, the last one ragged); loss and every
    gradient of the fused and the generic step against the oracle at the sweep tolerance, fp64-arbitrated. """
    from oracle import pinn_oracle as po
    import test_emu_engine as te
    net = dict(layout='faR fa fa+ R fa f+a f', features=[width] * 5 + [1], activation=['Tanh', 'Sigmoid', 'Tanh', 'Tanh', 'Sigmoid'])
    heavy = which.endswith('_any_activation')
    if heavy:       # the full breadth kernels (VAR 8 | 128) + pinn_wgrad_kernel<..., SKIPS, HEAVY>
        net = dict(layout='fRa fa f+a R f fa+ fa f', features=[width] * 6 + [1], activation=['Sin', 'SiLU', 'GELU', 'Softplus', 'Tanh'])
        which = which[:-len('_any_activation')]
    torch.manual_seed(width + len(which))
    eq_o, kw_o = te._layout_problems(po.D, torch, which, net)
    oracle = po.OracleSolver(eq_o, **kw_o)
    eq_p, kw = te._layout_problems(pa.D, torch, which, net)
    solver = pa.Solver(eq_p, **kw)
    assert solver.program is not None, solver.program_error
    load_params(solver, oracle.export_params())
    pts = np.random.RandomState(6).rand(1100, 2).astype(np.float32)
    loss_o = oracle.evaluate(pts)['loss']
    grads_o = oracle.export_grads()
    arbiter = {}

    def f64():
        if not arbiter:
            o64 = po.OracleSolver(eq_o, dtype=torch.float64, **kw_o)
            o64.import_params(oracle.export_params())
            arbiter['loss'] = o64.evaluate(pts)['loss']
            arbiter['grads'] = o64.export_grads()
        return arbiter
    lib = solver.model.net.lib
    for path in ('fused', 'generic'):
        solver.grads.zero_()
        xs_dev = torch.from_numpy(pts.copy()).cuda()
        if path == 'fused':
            solver._fused_step(xs_dev, 1)
        else:
            solver._generic_step(xs_dev, ('equation',), (), torch.nn.MSELoss(), 1)
        torch.cuda.synchronize()
        want_var = ('%d>' % (8 | 128),) if heavy else ('%d>' % (8 | 1024 | 128), '%d>' % (8 | 16 | 1024 | 128))
        assert lib.pinn_last_kernel_name().decode().rsplit(',', 1)[1] in want_var, lib.pinn_last_kernel_name()
        assert lib.pinn_last_wgrad_kernel_name().decode().endswith(',true,true>' if heavy else ',true,false>'), lib.pinn_last_wgrad_kernel_name()
        lay = solver.model.net.layout
        ok, err, arb = close_or_arbitrated([float(solver.grads[lay.off_loss])], [loss_o], lambda: [f64()['loss']], 1e-5, atol=0.0)
        record_margin('wide_residual_nets', (width, which, heavy, path), 'loss', err, 1e-5, arb)
        assert ok, (path, err)
        for i, (got, want) in enumerate(zip(export_grads(solver), grads_o)):
            if want is not None:
                ok, err, arb = close_or_arbitrated(got, want, lambda i=i: f64()['grads'][i], GRAD_RTOL)
                record_margin('wide_residual_nets', (width, which, heavy, path), 'gradient', err, GRAD_RTOL, arb)
                assert ok, (path, i, err)


@pytest.mark.parametrize('net', ['skip', 'two_skips', 'sin', 'identity', 'skip_to_top_wide', 'full64', 'softplus_silu_gelu',
                                 'nested_skips', 'nested_pre_activation', 'relu_family', 'softsign_gelutanh_mish', 'shrink_logsigmoid'])
@pytest.mark.parametrize('which', ['poisson', 'burgers'])
def test_layout_breadth_matches_the_oracle(pa, net, which):
    """ skip connections 'R ... +', per-layer activation lists, Sin, activation-free dense layers (reference
    model_torch.py:142-156): fused and generic Adam trajectories and predict against the oracle """
    from oracle import pinn_oracle as po
    import test_emu_engine as te
    eq_o, kw = te._layout_problems(po.D, torch, which, te.LAYOUTS[net])
    oracle = po.OracleSolver(eq_o, **kw)
    eq_p, kw = te._layout_problems(pa.D, torch, which, te.LAYOUTS[net])
    pts = np.random.RandomState(10).rand(3, 700, 2).astype(np.float32)
    start = oracle.export_params()
    oracle.fit(niters=3, batch_size=700, points=pts, lr=0.01)
    arb = {}
    for path in ('fused', 'generic'):
        solver = pa.Solver(eq_p, **kw)
        assert solver.program is not None, solver.program_error
        if path == 'generic':
            solver.program = None
        load_params(solver, start)
        solver.fit(niters=3, batch_size=700, sampler=FixedBatches(pts), lr=0.01)
        assert solver.last_fit_path == path
        np.testing.assert_allclose([float(v) for v in solver.losses], [float(v) for v in oracle.losses], rtol=3e-5)
        for i, (got, want) in enumerate(zip(export_params(solver), oracle.export_params())):
            # (three Adam steps of size lr = 0.01: from the second step on an entry moves by lr * f(g2 / g1), so the
            #  1e-4-level relative noise two fp32 implementations have on SMALL gradient entries of a second-order residual
            #  shows up as ~3e-6 absolute per entry, whatever the entry's own size -- hence the absolute term; a wrong
            #  update rule is off by lr * O(0.1 .. 1) = 1e-3 .. 1e-2. The losses above are the sharp check.
            #  Round 6: where the fp32 reference is the noisy side the fp64 trajectory arbitrates, like in the fuzz sweeps --
            #  `skip_to_top_wide` on the Poisson problem tripped the plain bound when the pre-pass constants stopped being
            #  rounded to fp32: a change of the source term in its last bits, amplified by Adam in entries with near-zero gradient)
            def o64(i=i):
                if 'o' not in arb:
                    o = po.OracleSolver(eq_o, dtype=torch.float64, **te._layout_problems(po.D, torch, which, te.LAYOUTS[net])[1])
                    o.import_params(start)
                    o.fit(niters=3, batch_size=700, points=pts, lr=0.01)
                    arb['o'] = o.export_params()
                return arb['o'][i]
            ok, err, arbitrated = close_or_arbitrated(got, want, o64, 3e-5, atol=2e-5, adam_move=3 * 0.01)
            record_margin('layout_breadth', (which, net, path, i), 'parameters', err, 3e-5, arbitrated)
            assert ok, (which, net, path, i, err)
    xs = [pts[0][:, i] for i in range(2)]
    assert np.abs(solver.predict(*xs) - oracle.predict(*xs)).max() < 2e-5


def test_deep_network_and_full_size_cfg5(pa):
    """ depth beyond the register-resident accumulators (8 hidden layers) and BASELINE config 5 (6x256) at a larger batch:
    loss and gradients against the oracle evaluated in chunks """
    from oracle import pinn_oracle as po
    kw = dict(ndims=2, boundary_condition=0.5, layout='fa' * 8 + 'f', features=[20] * 8 + [1], activation='Tanh')

    def eq(D):
        return lambda f, x, y: D(D(f, x), x) + D(D(f, y), y) - 5 * torch.sin(np.pi * (x + y))
    oracle = po.OracleSolver(eq(po.D), **kw)
    solver = pa.Solver(eq(pa.D), **kw)
    load_params(solver, oracle.export_params())
    pts = np.random.RandomState(12).rand(3, 500, 2).astype(np.float32)
    oracle.fit(niters=3, batch_size=500, points=pts, lr=0.01)
    solver.fit(niters=3, batch_size=500, sampler=FixedBatches(pts), lr=0.01)
    np.testing.assert_allclose([float(v) for v in solver.losses], [float(v) for v in oracle.losses], rtol=3e-5)

    torch.manual_seed(7)
    cfg, solver = make_solver('cfg5', pa)
    ocfg = pc.make_config('cfg5', po.D, torch)
    oracle = po.OracleSolver(ocfg['equation'], **ocfg['solver_kwargs'])
    oracle.import_params(export_params(solver))
    pts = pc.sample_points(cfg, 8192, seed=3)
    ev = oracle.evaluate(pts, chunk=2048)
    solver._fused_step(torch.from_numpy(pts).cuda(), 1)
    lay = solver.model.net.layout
    assert abs(float(solver.grads[lay.off_loss]) - ev['loss']) <= 1e-5 * ev['loss']
    for got, want in zip(export_grads(solver), oracle.export_grads()):
        assert grad_close(got, want)


def test_trainable_variable_constraint_and_freezing_on_the_gpu(pa):
    """ generic step path on the GPU (kernel streams -> the user's torch code with a trainable V -> kernel reverse sweep,
    constraint term through model(xs), freeze/unfreeze through the Adam mask): tutorial cells 50-60 vs the oracle """
    from test_emu_engine import _paired
    oracle, solver = _paired(pa, None)
    assert solver.program is not None and solver.residual_plan.n_vars == 1
    pts = np.random.RandomState(3).rand(8, 256, 1).astype(np.float32)
    terms = ['equation', 'constraint_0']
    solver.use_fused = False                        # the generic path is what this test pins
    oracle.fit(niters=4, batch_size=256, points=pts[:4], lr=0.05, loss_terms=terms)
    solver.fit(niters=4, batch_size=256, sampler=FixedBatches(pts[:4]), lr=0.05, loss_terms=terms)
    assert solver.last_fit_path == 'generic'
    np.testing.assert_allclose([float(v) for v in solver.losses], [float(v) for v in oracle.losses], rtol=5e-5)
    assert abs(float(solver.model.new_var) - float(oracle.model.new_var.detach())) < 2e-5
    oracle.model.new_var.requires_grad = False
    solver.model.freeze_trainable(variables=('new_var',))
    frozen = float(solver.model.new_var)
    solver.use_fused = True
    oracle.fit(niters=2, batch_size=256, points=pts[4:6], lr=0.05)
    solver.fit(niters=2, batch_size=256, sampler=FixedBatches(pts[4:6]), lr=0.05)
    assert float(solver.model.new_var) == frozen
    for got, want in zip(export_params(solver), oracle.export_params()):
        assert params_close(got, want, 1e-4)
    xs = np.linspace(0, 1, 9).astype(np.float32)
    assert np.abs(solver.predict(xs) - oracle.predict(xs)).max() < 2e-5


def test_trainable_variables_on_the_fused_path_on_the_gpu(pa):
    """ scalar V(...) as registers of the residual program: their gradients come out of the tile kernel (user slots of the
    gradient buffer) and the fused Adam launch updates them -- tutorial's ODE with a variable and a two-coefficient
    inverse problem (IC + BC, nonlinear term) vs the oracle, with ragged batches (tail tiles must not contribute) """
    from test_emu_engine import _paired, _inverse_problem
    from oracle import pinn_oracle as po
    oracle, solver = _paired(pa, None)
    pts = np.random.RandomState(5).rand(6, 1000, 1).astype(np.float32)
    oracle.fit(niters=6, batch_size=1000, points=pts, lr=0.05)
    solver.fit(niters=6, batch_size=1000, sampler=FixedBatches(pts), lr=0.05)
    assert solver.last_fit_path == 'fused'
    np.testing.assert_allclose([float(v) for v in solver.losses], [float(v) for v in oracle.losses], rtol=5e-5)
    assert abs(float(solver.model.new_var) - float(oracle.model.new_var.detach())) < 2e-5
    assert float(solver.model.new_var) != 1.0

    kw = dict(ndims=2, initial_condition=lambda x: torch.sin(np.pi * x), boundary_condition=0.0, layout='fafafaf',
              features=[64, 64, 64, 1], activation='Tanh')
    oracle = po.OracleSolver(_inverse_problem(po.D, po.V, torch), **kw)
    solver = pa.Solver(_inverse_problem(pa.D, pa.V, torch), **kw)
    load_params(solver, oracle.export_params())
    assert solver.program is not None and solver.residual_plan.n_vars == 2, solver.program_error
    pts = np.random.RandomState(6).rand(5, 4099, 2).astype(np.float32)
    oracle.fit(niters=5, batch_size=4099, points=pts, lr=0.02)
    solver.fit(niters=5, batch_size=4099, sampler=FixedBatches(pts), lr=0.02)
    assert solver.last_fit_path == 'fused'
    np.testing.assert_allclose([float(v) for v in solver.losses], [float(v) for v in oracle.losses], rtol=5e-5)
    for name in ('diffusivity', 'source'):
        assert abs(float(getattr(solver.model, name)) - float(getattr(oracle.model, name).detach())) < 2e-5
    for got, want in zip(export_params(solver), oracle.export_params()):
        assert params_close(got, want, 1e-4)


def test_constraint_terms_on_the_fused_path_on_the_gpu(pa):
    """ constraint loss terms (reference :451-457) as residual programs over the value stream on their own points, added to
    the equation's gradient by pinn_residual_step_add: tutorial cells 50-60 term by term, then two fit calls of a
    heat-type problem with a variable that only the constraint knows (born late in the reference: trap documented in
    DESIGN.md) """
    from test_emu_engine import _paired
    from oracle import pinn_oracle as po
    oracle, solver = _paired(pa, None)
    assert solver.constraint_plans[0] is not None, solver.constraint_errors
    pts = np.random.RandomState(8).rand(9, 1000, 1).astype(np.float32)
    for terms, lo in ((['equation', 'constraint_0'], 0), (['constraint_0'], 3), ('equation', 6)):
        oracle.fit(niters=3, batch_size=1000, points=pts[lo:lo + 3], lr=0.05, loss_terms=terms)
        solver.fit(niters=3, batch_size=1000, sampler=FixedBatches(pts[lo:lo + 3]), lr=0.05, loss_terms=terms)
        assert solver.last_fit_path == 'fused'
    np.testing.assert_allclose([float(v) for v in solver.losses], [float(v) for v in oracle.losses], rtol=5e-5)
    assert abs(float(solver.model.new_var) - float(oracle.model.new_var.detach())) < 2e-5

    def problem(D, V):
        def eq(u, x, t):
            return D(u, t) - 0.3 * D(D(u, x), x)

        def con(f, x, t):
            return f(np.array([0.25, 0.5, 0.75]), 0.5) - V('level', data=torch.Tensor([0.4])) * 2.0
        return eq, con
    kw = dict(ndims=2, initial_condition=lambda x: torch.sin(np.pi * x), boundary_condition=0.0, layout='fafafaf',
              features=[64, 64, 64, 1], activation='Tanh')
    eq_o, con_o = problem(po.D, po.V)
    eq_p, con_p = problem(pa.D, pa.V)
    oracle = po.OracleSolver(eq_o, constraints=con_o, **kw)
    solver = pa.Solver(eq_p, constraints=con_p, **kw)
    load_params(solver, oracle.export_params())
    pts = np.random.RandomState(9).rand(7, 4099, 2).astype(np.float32)
    terms = ['equation', 'constraint_0']
    oracle.fit(niters=4, batch_size=4099, points=pts[:4], lr=0.02, loss_terms=terms)
    solver.fit(niters=4, batch_size=4099, sampler=FixedBatches(pts[:4]), lr=0.02, loss_terms=terms)
    assert solver.last_fit_path == 'fused'
    assert float(solver.model.level) == float(oracle.model.level.detach()) == float(np.float32(0.4))
    oracle.fit(niters=3, batch_size=4099, points=pts[4:], lr=0.02, loss_terms=terms)
    solver.fit(niters=3, batch_size=4099, sampler=FixedBatches(pts[4:]), lr=0.02, loss_terms=terms)
    np.testing.assert_allclose([float(v) for v in solver.losses], [float(v) for v in oracle.losses], rtol=5e-5)
    assert abs(float(solver.model.level) - float(oracle.model.level.detach())) < 2e-5
    assert float(solver.model.level) != float(np.float32(0.4))
    for got, want in zip(export_params(solver), oracle.export_params()):
        assert params_close(got, want, 1e-4)


def test_default_sampler_trains_on_device(pa):
    """ the default sampler draws U[0,1)^d in HBM (reference model_torch.py:431 draws on the host); a short run of the
    README Poisson problem must bring the loss down and keep the hard boundary condition exact """
    torch.manual_seed(0)
    cfg, solver = make_solver('cfg1', pa)
    solver.fit(niters=600, batch_size=100, lr=0.005)
    losses = np.array([float(v) for v in solver.losses])
    assert len(losses) == 600 and losses[-50:].mean() < 0.05 * losses[:10].mean()
    edge = np.linspace(0, 1, 11).astype(np.float32)
    assert np.abs(solver.predict(edge, 0.0) - 1.0).max() < 1e-6 and np.abs(solver.predict(1.0, edge) - 1.0).max() < 1e-6


@pytest.mark.parametrize('case', ['poisson_3x64', 'ode_5x64', 'ode2_4x64', 'heat_4x64', 'poisson_4x64_sigmoid', 'poisson_2x64'])
def test_other_width_64_shapes_against_the_oracle(pa, case):
    """ the fast width-64 instantiations beyond the BASELINE shapes (and two shapes that fall to the generic kernels):
    one fused evaluation vs the oracle's nested autograd """
    from oracle import pinn_oracle as po

    def make(D):
        if case.startswith('poisson'):
            depth = {'poisson_3x64': 3, 'poisson_4x64_sigmoid': 4, 'poisson_2x64': 2}[case]
            act = 'Sigmoid' if 'sigmoid' in case else 'Tanh'
            eq = lambda f, x, y: D(D(f, x), x) + 2 * D(D(f, y), y) - torch.exp(-x) * torch.sin(np.pi * y)
            kw = dict(ndims=2, boundary_condition=0.3, layout='fa' * depth + 'f', features=[64] * depth + [1], activation=act)
        elif case == 'ode_5x64':
            eq = lambda f, x, e: D(f, x) - e * np.pi * torch.cos(e * np.pi * x)
            kw = dict(ndims=1, nparams=1, initial_condition=1.5, layout='fa' * 5 + 'f', features=[64] * 5 + [1], activation='Tanh')
        elif case == 'ode2_4x64':
            eq = lambda f, x: D(D(f, x), x) + 4 * f - torch.sin(3 * x)
            kw = dict(ndims=1, boundary_condition=0.0, layout='fa' * 4 + 'f', features=[64] * 4 + [1], activation='Tanh')
        else:
            eq = lambda f, x, y, t: D(D(f, x), x) + D(D(f, y), y) - 0.5 * D(f, t)
            kw = dict(ndims=3, boundary_condition=0, initial_condition=lambda x, y: x * y * (1 - x) * (1 - y),
                      layout='fa' * 4 + 'f', features=[64] * 4 + [1], activation='Tanh')
        return eq, kw
    eq_o, kw = make(po.D)
    torch.manual_seed(11)
    oracle = po.OracleSolver(eq_o, **kw)
    eq_p, kw = make(pa.D)
    solver = pa.Solver(eq_p, **kw)
    assert solver.program is not None, solver.program_error
    load_params(solver, oracle.export_params())
    d = solver.model.total
    pts = (np.random.RandomState(5).rand(2000, d) + (np.arange(d) >= kw['ndims'])).astype(np.float32)
    ev = oracle.evaluate(pts)
    solver._fused_step(torch.from_numpy(pts).cuda(), 1)
    lay = solver.model.net.layout
    assert abs(float(solver.grads[lay.off_loss]) - ev['loss']) <= 1e-5 * ev['loss']
    for got, want in zip(export_grads(solver), oracle.export_grads()):
        if want is not None:
            assert grad_close(got, want)


def test_data_parallel_step_path_with_a_one_rank_rccl_group(pa):
    """ SURVEY 8e / DESIGN 7: the N > 1 iteration -- tile kernel + reduction, RCCL all-reduce enqueued on the compute stream
    (pydens_amd/comm.py, ncclAllReduce called directly), ONE Adam launch that also records the loss -- run here with a
    world-size-1 `nccl` group (the only size a 1-GPU box offers): trajectory and parameters must equal the single-process
    fused path. """
    import os
    import torch.distributed as dist
    g = Golden('cfg4')
    _, ref = make_solver('cfg4', pa)
    load_params(ref, g.params)
    ref.fit(niters=len(g.losses), batch_size=g.points.shape[1], sampler=FixedBatches(g.points), lr=g.lr)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', RANK='0', WORLD_SIZE='1')
    dist.init_process_group('nccl', device_id=torch.device('cuda', torch.cuda.current_device()))
    try:
        _, solver = make_solver('cfg4', pa)
        load_params(solver, g.params)
        from pydens_amd.solver import FlatAdam
        solver.optimizer = FlatAdam(solver.model, lr=g.lr)
        solver.optimizer.refresh()
        solver.begin_data_parallel()
        assert solver._comm is not None and solver._comm.direct          # RCCL called directly, not through torch
        history = torch.zeros(len(g.losses), device='cuda')
        for it in range(len(g.losses)):
            xs = torch.from_numpy(g.points[it].copy()).cuda()
            solver._dp_step(xs, 1, loss_out=history.data_ptr() + 4 * it)
        solver.end_data_parallel()
    finally:
        dist.destroy_process_group()
    np.testing.assert_allclose(history.cpu().numpy(), np.array([float(v) for v in ref.losses]), rtol=1e-6)
    np.testing.assert_allclose(history.cpu().numpy(), g.losses, rtol=fit_rtol('cfg4'))
    for got, want in zip(export_params(solver), export_params(ref)):
        assert rel_l2(got, want) < 1e-6


@pytest.mark.parametrize('width', [128, 256])
def test_streamed_weight_gradients_chunk_by_chunk_on_the_gpu(pa, width):
    """ widths >= 128 (pinn_wgrad_kernel): a batch whose per-tile slabs exceed the budget goes through tile kernel ->
    weight-gradient kernel -> reduction chunk by chunk, gradients adding up; forced here with a tiny budget (one sweep of the
    persistent workgroups per chunk: 20 000 points = 1250 tiles -> 3 chunks at width 256, 5 at 128 with its two workgroups
    per CU) and compared with the one-pass result and with the oracle on a 4096-point prefix. """
    from oracle import pinn_oracle as po
    from pydens_amd import engine
    lib = engine.load_library()
    name = 'cfg3' if width == 128 else 'cfg5'
    cfg, solver = make_solver(name, pa)
    g = Golden(name)
    load_params(solver, g.params)
    pts = pc.sample_points(cfg, 20000, seed=5)
    xs = torch.from_numpy(pts).cuda()
    solver._fused_step(xs, 1)
    one_pass = solver.grads.clone()
    try:
        lib.pinn_debug_wgx_chunk_bytes(solver.model.net.handle, 1)
        solver.model._workspaces.clear()
        solver._fused_step(xs, 1)
    finally:
        lib.pinn_debug_wgx_chunk_bytes(solver.model.net.handle, 0)
        solver.model._workspaces.clear()
    lay = solver.model.net.layout
    assert rel_l2(solver.grads[:lay.p_core].cpu().numpy(), one_pass[:lay.p_core].cpu().numpy()) < 2e-6
    ocfg = pc.make_config(name, po.D, torch)
    oracle = po.OracleSolver(ocfg['equation'], **ocfg['solver_kwargs'])
    oracle.import_params(g.params)
    ev = oracle.evaluate(pts[:4096], chunk=1024)
    solver._fused_step(xs[:4096].contiguous(), 1)
    assert abs(float(solver.grads[lay.off_loss]) - ev['loss']) <= 1e-5 * ev['loss']
    for got, want in zip(export_grads(solver), oracle.export_grads()):
        if want is not None:
            assert grad_close(got, want)


def test_seeded_numpy_sampler_keys_the_device_sampler(pa):
    """ `NumpySampler(..., seed=k)` (reference model_torch.py:433 draws from that seeded generator): the batches depend on
    the sampler's seed alone -- two solvers under different torch seeds see the same points, another seed gives others. """
    def run(torch_seed, sampler_seed):
        torch.manual_seed(torch_seed)
        _, solver = make_solver('cfg4', pa)
        sampler = pa.NumpySampler('uniform', seed=sampler_seed) & pa.NumpySampler('uniform', low=1, high=5, seed=sampler_seed + 1)
        return solver._sample(1000, sampler).cpu().numpy(), solver._sample(1000, sampler).cpu().numpy()
    a0, a1 = run(1, 7)
    b0, b1 = run(2, 7)
    c0, _ = run(1, 8)
    assert np.array_equal(a0, b0) and np.array_equal(a1, b1) and not np.array_equal(a0, a1) and not np.array_equal(a0, c0)
    assert a0[:, 1].min() >= 1 and a0[:, 1].max() < 5
    # unseeded samplers follow torch's generator instead (like the reference's default torch.rand columns)
    torch.manual_seed(5); _, s1 = make_solver('cfg4', pa); x1 = s1._sample(100, None).cpu().numpy()
    torch.manual_seed(5); _, s2 = make_solver('cfg4', pa); x2 = s2._sample(100, None).cpu().numpy()
    torch.manual_seed(6); _, s3 = make_solver('cfg4', pa); x3 = s3._sample(100, None).cpu().numpy()
    assert np.array_equal(x1, x2) and not np.array_equal(x1, x3)


def test_closure_constants_and_reassigned_equations_take_effect_in_the_next_fit(pa):
    """ the reference calls equation(u_hat, *xs) in every iteration (model_torch.py:448); here the callable was lowered
    once, so every fit call re-checks the lowering against the live callable """
    from oracle import pinn_oracle as po
    coef = {'k': 1.0}

    def problem(D):
        def pde(f, x, y):
            return D(D(f, x), x) + D(D(f, y), y) - coef['k'] * torch.sin(np.pi * (x + y))
        return pde, dict(ndims=2, boundary_condition=1, layout='fa fa f', features=[16, 16, 1], activation='Tanh')
    eq_o, kw = problem(po.D)
    oracle = po.OracleSolver(eq_o, **kw)
    eq_p, kw = problem(pa.D)
    solver = pa.Solver(eq_p, **kw)
    load_params(solver, oracle.export_params())
    pts = np.random.RandomState(2).rand(4, 500, 2).astype(np.float32)
    for k, sl in ((1.0, slice(0, 2)), (5.0, slice(2, 4))):
        coef['k'] = k
        oracle.fit(niters=2, batch_size=500, points=pts[sl], lr=0.01)
        solver.fit(niters=2, batch_size=500, sampler=FixedBatches(pts[sl]), lr=0.01)
        assert solver.last_fit_path == 'fused'
    np.testing.assert_allclose([float(v) for v in solver.losses], [float(v) for v in oracle.losses], rtol=3e-5)


def test_convblockmodel_subclass_as_model_plugin(pa):
    """ the reference's plug-in seam `Solver(model=...)` (model_torch.py:299-313): a subclass that sets up the fully
    connected net its own way runs on the kernels; a subclass whose forward() is not built on self.conv_block is refused loudly
    (forward() with torch code AROUND the network: test_model_subclass_with_its_own_forward_on_the_gpu). """
    class MyNet(pa.ConvBlockModel):
        def __init__(self, **kwargs):
            kwargs.setdefault('layout', 'fa fa f')
            kwargs.setdefault('features', [24, 24, 1])
            kwargs.setdefault('activation', 'Tanh')
            super().__init__(**kwargs)

    class Custom(pa.ConvBlockModel):
        def forward(self, xs):
            return xs.sum(dim=1, keepdim=True)

    def pde(f, x):
        return pa.D(f, x) - 2 * np.pi * torch.cos(2 * np.pi * x)
    solver = pa.Solver(pde, ndims=1, initial_condition=0.5, model=MyNet)
    assert isinstance(solver.model, MyNet) and solver.model.layer_dims == [1, 24, 24, 1]
    solver.fit(niters=50, batch_size=256, lr=0.01)
    assert solver.last_fit_path == 'fused' and float(solver.losses[-1]) < float(solver.losses[0])
    with pytest.raises(NotImplementedError):
        pa.Solver(pde, ndims=1, initial_condition=0.5, model=Custom)


@pytest.mark.parametrize('gemm', ['fp32', 'bf16x3'])
def test_full_size_cfg4_against_the_chunked_oracle(pa, gemm):
    """ BASELINE config 4 at the per-GPU batch of its 8-GPU step (131 072 points): loss and every gradient tensor against the
    oracle evaluated in chunks (S = 2 streams: cheap enough for a test run) -- beside the prefix + shard-sum properties the
    1 048 576-point case is tied to the oracle with. """
    from oracle import pinn_oracle as po
    torch.manual_seed(5)
    cfg, solver = make_solver('cfg4', pa, gemm=gemm)
    ocfg = pc.make_config('cfg4', po.D, torch)
    oracle = po.OracleSolver(ocfg['equation'], **ocfg['solver_kwargs'])
    oracle.import_params(export_params(solver))
    pts = pc.sample_points(cfg, 131072, seed=2)
    ev = oracle.evaluate(pts, chunk=32768)
    solver._fused_step(torch.from_numpy(pts).cuda(), 1)
    lay = solver.model.net.layout
    assert abs(float(solver.grads[lay.off_loss]) - ev['loss']) <= 1e-5 * ev['loss']
    for got, want in zip(export_grads(solver), oracle.export_grads()):
        if want is not None:
            assert grad_close(got, want)


_BENCH_ORACLE_CACHE = {}


@pytest.mark.parametrize('gemm', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('name', ['cfg1', 'cfg2', 'cfg3', 'cfg4', 'cfg5'])
def test_bench_parity_rule_on_every_baseline_config(pa, name, gemm):
    """ the rule bench.py applies to its own line (`parity_checked`: loss and every gradient tensor of the step path against the oracle,
    |ours - f64| <= max(2 |ref32 - f64|, 1e-5 |f64|) -- SURVEY 8c item 5) on all five BASELINE configs, both GEMM modes, three point
    seeds. Round 5's committed config-4 lines carried ok = false (d loss / d b_L 1.15e-5 from fp64 at seed 99) and no test said so: the
    x-only source term e pi cos(e pi x) evaluated in fp32 -- by the reference and by the kernels alike -- carries a systematic error
    (the rounded pi shifts the cosine's argument the same way in every point) that survives the cancelling batch sum; the pre-pass
    runs in fp64 since (include/pinn.h pre_consts64, tools/cfg4_bl_probe.py). """
    import bench
    torch.manual_seed(0)
    cfg, solver = make_solver(name, pa, gemm=gemm)
    for seed in (99, 100, 101):
        res = bench.parity_check(name, solver, False, None, n_points=100 if name == 'cfg1' else 4096, seed=seed, cache=_BENCH_ORACLE_CACHE)
        record_margin('bench_parity_rule', (name, gemm, seed), 'grad', res['worst_gradient_rel_err'], 1e-5, True)
        assert res['ok'], (name, gemm, seed, res)


def test_known_answers_of_the_tutorial(pa):
    """ the analytic solutions the reference's tutorial plots its approximations against
    (tutorials/1. Solving PDEs.ipynb cells 12-16, 28-34, 50-63), trained with the tutorial's own settings on the device:
    ODE f' = 2 pi cos(2 pi x), f(0) = 1/2 -> sin(2 pi x) + 1/2; the parametric family f' = e pi cos(e pi x), f(0) = 2 ->
    sin(e pi x) + 2; the inverse problem f' = 2 pi cos(2 pi x) - V, f(0) = 1, constraint f(1/2) = 0 -> V = 2 and
    f = sin(2 pi x) + 1 - 2 x. """
    xs = np.linspace(0, 1, 100).astype(np.float32)
    torch.manual_seed(3)
    solver = pa.Solver(lambda f, x: pa.D(f, x) - 2 * np.pi * torch.cos(2 * np.pi * x), ndims=1, initial_condition=.5,
                       activation='Tanh', layout='fafaf', features=[12, 10, 1])
    solver.fit(niters=500, batch_size=400, lr=0.02)
    err = np.abs(solver.predict(xs)[:, 0] - (np.sin(2 * np.pi * xs) + .5)).max()
    assert solver.last_fit_path == 'fused' and err < 0.02, err          # (the reference's own step reaches 0.002)

    torch.manual_seed(3)
    solver = pa.Solver(lambda f, x, e: pa.D(f, x) - e * np.pi * torch.cos(e * np.pi * x), ndims=1, initial_condition=2.0,
                       nparams=1)                                   # default net: 'fafaf' [20, 30, 1] Sigmoid
    sampler = pa.NS('u') & pa.NS('u', low=.5, high=5.5)
    solver.fit(niters=7000, batch_size=700, sampler=sampler, lr=0.01)
    for eps in (1.0, 2.5, 4.0):
        err = np.abs(solver.predict(xs, eps)[:, 0] - (np.sin(eps * np.pi * xs) + 2)).max()
        assert err < 0.08, (eps, err)                                # (reference: 0.013 .. 0.018)

    torch.manual_seed(3)

    def odevar(f, x):
        return pa.D(f, x) - 2 * np.pi * torch.cos(2 * np.pi * x) + pa.V('new_var', data=torch.Tensor([1.0]))
    solver = pa.Solver(odevar, ndims=1, initial_condition=1, constraints=lambda f, x: f(torch.tensor([0.5])))
    solver.model.freeze_trainable(variables=('new_var',))
    solver.fit(niters=200, batch_size=500, lr=0.1)
    assert float(solver.model.new_var) == 1.0
    err = np.abs(solver.predict(xs)[:, 0] - (np.sin(2 * np.pi * xs) + 1 - xs)).max()      # V frozen at 1
    assert err < 0.15, err                                          # (reference: 0.02 .. 0.05)
    solver.model.unfreeze_trainable(variables=['new_var'])
    for _ in range(4):                                              # the tutorial's cell 60, repeated until V settles
        solver.fit(niters=100, batch_size=100, lr=0.1, loss_terms=['equation', 'constraint_0'])
    assert solver.last_fit_path == 'fused'
    v = float(solver.model.new_var)
    err = np.abs(solver.predict(xs)[:, 0] - (np.sin(2 * np.pi * xs) + 1 - 2 * xs)).max()
    assert abs(v - 2.0) < 0.05 and err < 0.1, (v, err)              # (reference: V = 1.997 .. 2.004, error 0.006 .. 0.02)


@pytest.mark.parametrize('problem', ['mixed', 'composite', 'D_of_mixed'])
def test_mixed_partial_and_composite_D_generic_path_on_the_gpu(pa, problem):
    import test_emu_engine as te
    te._generic_D_case(pa, problem, {})


@pytest.mark.parametrize('which', ['scaled_ansatz', 'no_ansatz_head', 'with_constraint', 'normalised_inputs', 'normalised_mixed',
                                   'map_sine', 'map_mixing', 'map_time_warp', 'map_mixing_third'])
def test_model_subclass_with_its_own_forward_on_the_gpu(pa, which):
    import test_emu_engine as te
    te._custom_forward_case(pa, which, {})


def test_twenty_hidden_layers_on_the_gpu(pa):
    """ deeper than one 64-bit word of activation codes (PINN_MAX_LAYERS 32): per-layer activations and a residual block at the top """
    import test_emu_engine as te
    te._deep_network_case(pa, 20, {})


@pytest.mark.parametrize('which', ['reaction_2d', 'allen_cahn', 'not_combinable'])
def test_residual_programs_on_one_combined_stream_on_the_gpu(pa, which):
    import test_emu_engine as te
    te._combined_program_case(pa, which, {})


def test_prepass_with_more_registers_than_a_full_sweep_holds_on_the_gpu(pa):
    import test_emu_engine as te
    te._wide_prepass_case(pa, {})


@pytest.mark.parametrize('name', ['nested_acts', 'mixed3', 'biharm', 'act_params', 'mixed31', 'mixed111'])
def test_breadth_features_match_reference_golden_on_the_gpu(pa, name):
    """ round 5 breadth (nested skips + second-set activations, mixed third order, fourth order) against the fixtures generated from the
    unmodified reference: predict, loss, gradients, K-step trajectory (tests/test_golden_extras.py holds the case) """
    import test_golden_extras as tg
    tg.golden_extra_case(pa, name, {}, test='gpu_golden_extra')


@pytest.mark.parametrize('which', ['beam_1d', 'kuramoto_sivashinsky', 'time_fourth', 'biharmonic', 'any_activation', 'beam_wide', 'beam_wide_sin', 'xxxt_gate',
                                   'tttp_gate'])
def test_fourth_order_streams_on_the_gpu(pa, which):
    import test_emu_engine as te
    te._fourth_order_case(pa, which, {})


@pytest.mark.parametrize('which', ['two_third_order_columns', 'third_beside_second', 'mixed_third_space', 'mixed_third_time', 'mixed_third_both', 'three_columns_time'])
def test_third_order_direction_groups_on_the_gpu(pa, which):
    """ equations with more third-order content than one kernel call carries: generic path, one call per third-order column """
    import test_emu_engine as te
    te._direction_groups_case(pa, which, {})


@pytest.mark.parametrize('which', ['ode_space', 'ode_time', 'wide_wgx', 'sin_skip', 'gelu'])
def test_third_order_streams_on_the_gpu(pa, which):
    """ u_xxx-type equations: third-order jets, ansatz product rules and reverse sweep on the device (fused and generic
    paths, widths 32 and 128 -- the latter through the streamed weight-gradient kernel), arbitrated by the fp64 oracle: three
    nested fp32 autograd sweeps of the reference's own arithmetic are off by up to 1.5e-4 here """
    from oracle import pinn_oracle as po
    import test_emu_engine as te
    eq_o, kw = te._third_order_problems(po.D, torch, which)
    oracle32 = po.OracleSolver(eq_o, **kw)
    oracle = po.OracleSolver(eq_o, dtype=torch.float64, **kw)
    start = oracle32.export_params()
    oracle.import_params(start)
    d, n = kw['ndims'], 700
    pts = np.random.RandomState(7).rand(3, n, d).astype(np.float32)
    if which == 'ode_time':
        pts = 0.5 + 1.5 * pts
    ev32, g32 = oracle32.evaluate(pts[0]), oracle32.export_grads()
    ev, g_want = oracle.evaluate(pts[0]), oracle.export_grads()
    oracle.fit(niters=3, batch_size=n, points=pts, lr=0.01)
    for path in ('fused', 'generic'):
        eq_p, kw = te._third_order_problems(pa.D, torch, which)
        solver = pa.Solver(eq_p, **kw)
        assert solver.spec.n3 == 1 and solver.program is not None, solver.program_error
        load_params(solver, start)
        if path == 'generic':
            solver.program = None
        else:
            solver._fused_step(torch.from_numpy(pts[0].copy()).cuda(), 1)
            lay = solver.model.net.layout
            loss = float(solver.grads[lay.off_loss])
            assert abs(loss - ev['loss']) <= max(2 * abs(ev32['loss'] - ev['loss']), 1e-5 * ev['loss'])
            for got, want, w32 in zip(export_grads(solver), g_want, g32):
                if want is not None:
                    err = np.linalg.norm(np.asarray(got, dtype=np.float64) - want)
                    assert err <= max(2 * np.linalg.norm(np.asarray(w32, dtype=np.float64) - want), 1e-4 * np.linalg.norm(want))
        solver.fit(niters=3, batch_size=n, sampler=FixedBatches(pts), lr=0.01)
        assert solver.last_fit_path == path
        np.testing.assert_allclose([float(v) for v in solver.losses], [float(v) for v in oracle.losses], rtol=1e-4)
        for got, want in zip(export_params(solver), oracle.export_params()):
            assert params_close(got, want, 1e-4, atol=2e-5)


def _two_rank_worker(rank, world, port, out_dir):
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    import pydens_amd as pa2
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', device_id=torch.device('cuda', rank))
    g = Golden('cfg4')
    _, solver = make_solver('cfg4', pa2, device=torch.device('cuda', rank))
    if rank == 0:
        load_params(solver, g.params)               # the other rank starts elsewhere: fit broadcasts from rank 0
    shard = g.points[:, rank::world]
    solver.fit(niters=len(g.losses), batch_size=g.points.shape[1], sampler=FixedBatches(shard), lr=g.lr)
    info = solver._comm.describe()
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), losses=np.array([float(v) for v in solver.losses]),
             direct=np.array([1 if solver._comm.direct else 0]), n_ranks=np.array([info['n_ranks']]),
             **{f'p{i}': p for i, p in enumerate(export_params(solver))})
    solver.end_data_parallel()
    dist.destroy_process_group()


def test_two_ranks_over_rccl_follow_the_single_process_trajectory(pa):
    """ the direct-RCCL branch of pydens_amd/comm.py (ncclCommInitRank over a broadcast ncclUniqueId,
    ncclAllReduce on the compute stream) with MORE than one rank -- runs wherever two devices are visible (the driver's
    multi-GPU node), skipped on a one-GPU box. Two ranks on half batches must follow the golden single-process trajectory. """
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two HIP devices')
    import socket
    import tempfile
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    g = Golden('cfg4')
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_two_rank_worker, args=(2, port, tmp), nprocs=2, join=True)
        for rank in range(2):
            z = np.load(os.path.join(tmp, f'rank{rank}.npz'))
            assert int(z['direct'][0]) == 1 and int(z['n_ranks'][0]) == 2       # RCCL itself, and it saw both ranks
            np.testing.assert_allclose(z['losses'], g.losses, rtol=fit_rtol('cfg4'))
            for i, want in enumerate(g.finals):
                assert rel_l2(z[f'p{i}'], want) < fit_rtol('cfg4'), (rank, i)


_FULL_BATCH_ORACLE = {}


@pytest.mark.parametrize('gemm', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('name,n', [('cfg3', 262144), ('cfg5', 131072)])
def test_wide_nets_at_their_full_batch_against_the_chunked_oracle(pa, name, n, gemm):
    """ BASELINE config 3 at its full batch (262 144 points) and config 5 at the per-GPU batch of its 8-GPU step
    (131 072 points) -- the streamed weight-gradient path -- against the oracle evaluated in chunks of 16 384 points: loss and every
    gradient tensor, not only a prefix plus shard / permutation invariants. """
    from oracle import pinn_oracle as po
    torch.manual_seed(5)
    cfg, solver = make_solver(name, pa, gemm=gemm)
    pts = pc.sample_points(cfg, n, seed=2)
    start = export_params(solver)
    cached = _FULL_BATCH_ORACLE.get((name, n))
    if cached is None or any(not np.array_equal(a, b) for a, b in zip(cached[0], start)):
        # (the oracle's pass over 131 072 - 262 144 points takes 20 - 150 s of host time: both GEMM modes start from the same
        #  parameters and points, so it is evaluated once)
        ocfg = pc.make_config(name, po.D, torch)
        oracle = po.OracleSolver(ocfg['equation'], **ocfg['solver_kwargs'])
        oracle.import_params(start)
        ev = oracle.evaluate(pts, chunk=16384)
        cached = _FULL_BATCH_ORACLE[(name, n)] = ([np.array(a, copy=True) for a in start], ev['loss'], oracle.export_grads())
    _, loss_o, grads_o = cached
    solver._fused_step(torch.from_numpy(pts).cuda(), 1)
    lay = solver.model.net.layout
    assert abs(float(solver.grads[lay.off_loss]) - loss_o) <= 1e-5 * loss_o
    for got, want in zip(export_grads(solver), grads_o):
        if want is not None:
            assert grad_close(got, want)


@pytest.mark.parametrize('gemm', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('name,n', [('cfg2', 16384), ('cfg2', 65536), ('cfg4', 131072), ('cfg3', 65536), ('cfg3', 262144), ('cfg5', 32768),
                                    ('cfg5', 131072)])
def test_baseline_kernels_are_bitwise_repeatable(pa, name, n, gemm):
    """ the same step eight times gives the same bits (fixed summation order, no atomics), exact-fp32 and split-bf16 kernels at the
    BASELINE batches. A guard with a history: experiment builds whose two waves per SIMD run OUT OF STEP (two independent workgroups per
    CU, flag-synchronised teams) return run-to-run different gradients when hipcc's SLP vectoriser has put packed fp32 code beside the
    bf16 MFMAs ("run-to-run different gradients"; tools/var2.sh reproduces it); the shipped kernels keep the
    two waves of a SIMD in one workgroup, phase by phase behind the same barriers """
    torch.manual_seed(3)
    cfg, solver = make_solver(name, pa, gemm=gemm)
    pts = torch.from_numpy(pc.sample_points(cfg, n, seed=3)).cuda()
    solver._fused_step(pts, 1)
    assert ran_split_kernel(solver) == (gemm == 'bf16x3')
    first = solver.grads.clone()
    for _ in range(7):
        solver._fused_step(pts, 1)
        assert torch.equal(solver.grads, first)


@pytest.mark.parametrize('name,n,launches', [('cfg2', 65536, 1000), ('cfg4', 131072, 1000), ('cfg3', 262144, 150), ('cfg5', 131072, 300)])
def test_split_kernels_soak_a_thousand_launches_bitwise(pa, name, n, launches):
    """ round 5: the split-bf16 kernels at their full BASELINE batches, launch after launch, every gradient buffer
    compared bit for bit with the first one ON THE DEVICE (one flag, read once at the end). The hazard round 4 found -- a packed fp32
    instruction reading a source the next instruction overwrites, beside a SIMD partner in a bf16-MFMA phase -- showed in ~6 % of the
    workgroups of a launch when it was live; the build now takes these units through asm_guard.py (no such overwrite within three issue
    slots, tests/test_asm_guard.py) and this soak is the watch behind the guard. """
    torch.manual_seed(5)
    cfg, solver = make_solver(name, pa, gemm='bf16x3')
    pts = torch.from_numpy(pc.sample_points(cfg, n, seed=4)).cuda()
    solver._fused_step(pts, 1)
    assert ran_split_kernel(solver)
    first = solver.grads.clone()
    differing = torch.zeros((), dtype=torch.int64, device=first.device)
    for _ in range(launches):
        solver._fused_step(pts, 1)
        differing += (solver.grads.view(torch.int32) != first.view(torch.int32)).any().to(torch.int64)
    assert int(differing.item()) == 0, f'{int(differing.item())} of {launches} launches differ from the first one'


def test_accurate_tanh_mode_on_a_trained_state(pa):
    """ Solver.set_tanh_mode('accurate') (round 5): the Poisson-box kernel of config 2 with the polynomial tanh below |z| = 0.45 -- same
    golden parity as the default form, and on a (partly) trained state a gradient error against the fp64 oracle no worse than 1.4x the fp32
    reference's own (the default form: up to ~1.9x; SURVEY 8c item 5 allows 2x). """
    from oracle import pinn_oracle as po
    g = Golden('cfg2')
    cfg, solver = make_solver('cfg2', pa)
    solver.set_tanh_mode('accurate')
    load_params(solver, g.params)
    xs = torch.from_numpy(g.points[0].astype(np.float32)).cuda()
    solver._fused_step(xs, 1)
    assert solver.model.net.lib.pinn_last_kernel_name().decode().split(',')[5] == str(0x100)
    assert abs(float(solver.grads[solver.model.net.layout.off_loss]) - g.loss0) <= 1e-5 * g.loss0
    for got, want in zip(export_grads(solver), g.grads):
        if want is not None:
            assert grad_close(got, want)
    # a trained state: 300 Adam steps, then ours / ref32 against fp64 on fresh points
    torch.manual_seed(6)
    solver.fit(niters=300, batch_size=4096, lr=0.005)
    params = export_params(solver)
    pts = pc.sample_points(cfg, 4096, seed=77)
    ocfg = pc.make_config('cfg2', po.D, torch)
    refs = {}
    for dtype in (torch.float32, torch.float64):
        oracle = po.OracleSolver(ocfg['equation'], dtype=dtype, **ocfg['solver_kwargs'])
        oracle.import_params(params)
        oracle.evaluate(pts, chunk=2048)
        refs[dtype] = oracle.export_grads()
    solver.grads.zero_()
    solver._fused_step(torch.from_numpy(pts).cuda(), 1)
    ratios = []
    for got, a32, a64 in zip(export_grads(solver), refs[torch.float32], refs[torch.float64]):
        if a64 is not None:
            a64 = np.asarray(a64, dtype=np.float64)
            ratios.append(np.linalg.norm(np.asarray(got, dtype=np.float64) - a64) / max(np.linalg.norm(np.asarray(a32, dtype=np.float64) - a64), 1e-30))
    print('accurate tanh: gradient error vs fp64, ours / fp32 reference, per tensor:', [round(float(r), 2) for r in ratios])
    assert max(ratios) <= 1.4, ratios


@pytest.mark.parametrize('which', ['heat', 'wave', 'burgers'])
def test_evolution_shape_in_x_t_on_the_two_team_kernels(pa, which):
    import test_emu_engine as te
    te._evolution_case(pa, which, {}, 3000)


@pytest.mark.parametrize('which', ['ode_fused', 'poisson_groups', 'heat_sigmoid', 'advection_breadth'])
def test_hidden_width_512_on_the_gpu(pa, which):
    """ round 6: hidden widths 257 .. 512 (S <= 3 streams per kernel call, direction groups beyond, block passes of the weight-gradient kernel) """
    import test_emu_engine as te
    te._wide512_case(pa, which, {}, 4000)


@pytest.mark.parametrize('hp', [128, 256])
def test_wide_sin_nets_on_the_static_activation_kernel(pa, hp):
    import test_emu_engine as te
    te._wide_sin_case(pa, hp, {}, 3000)


def test_sin_net_of_depth_four_on_the_static_kernel(pa):
    """ the 4 x 64 'Sin' breadth workload of bench.py (static-depth kernel with the Dirichlet-box facts fixed) against the oracle """
    from oracle import pinn_oracle as po
    torch.manual_seed(4)
    ocfg = pc.make_config('sin64', po.D, torch)
    oracle = po.OracleSolver(ocfg['equation'], **ocfg['solver_kwargs'])
    cfg, solver = make_solver('sin64', pa)
    load_params(solver, oracle.export_params())
    pts = pc.sample_points(cfg, 3000, seed=9, steps=2)
    oracle.fit(niters=2, batch_size=3000, points=pts, lr=0.005)
    solver.fit(niters=2, batch_size=3000, sampler=FixedBatches(pts), lr=0.005)
    assert solver.model.net.lib.pinn_last_kernel_name().decode() == 'pinn_tile_kernel<64,2,1,1,3,2,true,272>'      # (round 6: two teams of 16-point tiles)
    np.testing.assert_allclose([float(v) for v in solver.losses], [float(v) for v in oracle.losses], rtol=2e-5)
    for got, want in zip(export_params(solver), oracle.export_params()):
        assert params_close(got, want, 2e-5)


@pytest.mark.parametrize('name,batch', [('cfg1', 100), ('cfg4', 2048)])
def test_fit_chunks_as_launch_graphs_follow_the_eager_loop_bit_for_bit(pa, name, batch, monkeypatch):
    """ Solver.fit at small batches replays every 128-iteration chunk as ONE launch graph (pinn_fit_steps_graph: the Philox batch
    counter, the Adam step with its bias corrections and the loss slot come from a device control block); the trajectory -- every loss,
    every parameter, the Adam state -- must be the eager loop's, bit for bit, across several chunks and a second fit that continues """
    def run(graph):
        monkeypatch.setenv('PYDENS_AMD_FIT_GRAPH', '1' if graph else '0')
        monkeypatch.setenv('PYDENS_AMD_FIT_PERSIST', '0')           # (this test is about the launch-graph replay: same kernels, same bits)
        torch.manual_seed(21)
        cfg, solver = make_solver(name, pa)
        sampler = (pa.NumpySampler('uniform') & pa.NumpySampler('uniform', low=1, high=5)) if name == 'cfg4' else None
        solver.fit(niters=300, batch_size=batch, sampler=sampler, lr=0.005)             # eager chunk + capture, replay, short tail (eager)
        solver.fit(niters=260, batch_size=batch, sampler=sampler, lr=0.005, optimizer=None)      # continues: the graph is replayed at once
        assert solver.last_fit_path == 'fused'
        return (np.array([float(v) for v in solver.losses]), solver.model.flat.detach().cpu().numpy().copy(),
                solver.optimizer.exp_avg.cpu().numpy().copy(), int(solver.optimizer.step_count.item()))
    l0, p0, m0, t0 = run(False)
    l1, p1, m1, t1 = run(True)
    assert t0 == t1 == 560
    assert np.array_equal(l0, l1)
    assert np.array_equal(p0, p1) and np.array_equal(m0, m1)


@pytest.mark.parametrize('mode,which', [(2, 'cfg1'), (2, 'ode_16'), (2, 'ode_default_net'), (2, 'program_with_variable'),
                                        (2, 'skip_sin'), (1, 'cfg1'), (1, 'ode_default_net'), (1, 'program_with_variable')])
def test_fit_chunk_as_one_launch_follows_the_eager_loop(pa, mode, which, monkeypatch):
    """ round 5: narrow nets at the reference's batch sizes run a whole chunk of fit iterations -- sampling, tile body,
    the sum of the partial rows, Adam -- in ONE launch (pinn_fit_kernel.h). Mode 2 (default for batches of a few tiles): ONE hardware
    workgroup of up to eight virtual workgroups on one CU, a workgroup barrier per iteration. Mode 1 (opt-in, measured slower than
    launch-graph replay, profiles/r05_small_fit_rate.txt): the workgroups of a grid meet once per iteration in a device-scope arrive /
    wait. Same Adam scalars and Philox counters as the eager loop: every loss, every parameter, the Adam moments and the step counter
    follow it to fp32 round-off, over whole chunks, a short tail and a second fit that continues. """
    import test_emu_engine as te
    te._one_launch_case(pa, which, {}, monkeypatch, mode, (300, 130), pa.engine.load_library())


@pytest.mark.parametrize('which', ['cfg2_forced_generic', 'tensor_variable'])
def test_generic_fit_as_a_launch_graph_follows_the_eager_loop_bit_for_bit(pa, which, monkeypatch):
    """ the generic step (pinn_jet_forward -> the user's torch code + autograd -> pinn_jet_backward) is recorded after three eager
    iterations and replayed as ONE launch graph (Solver._generic_step_auto); same kernels in the same order: every loss, every
    parameter and the Adam state equal the eager loop's bit for bit; a second fit records anew. """
    def run(graph):
        monkeypatch.setenv('PYDENS_AMD_STEP_GRAPH', '1' if graph else '0')
        monkeypatch.setattr(pa.Solver, 'GENERIC_GRAPH_MIN_REPLAYS', 0)      # (these fits are short on purpose: record whatever is left)
        torch.manual_seed(22)
        if which == 'cfg2_forced_generic':
            cfg, solver = make_solver('cfg2', pa)
            solver.program = None
            batch = 4096
        else:
            def eq(f, x, t):                 # a vector-valued trainable: the tracer leaves it to torch autograd (generic path)
                w = pa.V('w', data=torch.Tensor([0.5, 1.5]))
                return pa.D(f, t) - 0.1 * w[0] * pa.D(pa.D(f, x), x) + w[1] * f * pa.D(f, x)
            solver = pa.Solver(eq, ndims=2, boundary_condition=0.0, initial_condition=0.3, layout='fa fa f', features=[32, 32, 1],
                               activation='Tanh')
            batch = 1000
        solver.fit(niters=20, batch_size=batch, lr=0.005)
        replays = (getattr(solver, '_generic_graph', None) or {}).get('replays', 0)
        solver.fit(niters=12, batch_size=batch, lr=0.005, optimizer=None)
        assert solver.last_fit_path == 'generic'
        replays += (getattr(solver, '_generic_graph', None) or {}).get('replays', 0)
        return (np.array([float(v) for v in solver.losses]), solver.model.flat.detach().cpu().numpy().copy(),
                solver.optimizer.exp_avg.cpu().numpy().copy(), replays, (getattr(solver, '_generic_graph', None) or {}).get('error'))
    l0, p0, m0, r0, _ = run(False)
    l1, p1, m1, r1, err = run(True)
    assert r0 == 0 and r1 == (20 - 3) + (12 - 3), (r1, err)
    assert np.array_equal(l0, l1)
    assert np.array_equal(p0, p1) and np.array_equal(m0, m1)


def test_generic_terms_with_a_constraint_as_a_launch_graph_follow_the_eager_loop_bit_for_bit(pa, monkeypatch):
    """ equation + constraint terms on the GENERIC path (the tutorial's variable + constraint problem, cells 50-60, with
    use_fused = False: kernel streams, the equation and the constraint -- the model on a fixed point -- in torch): recorded after three
    eager iterations and replayed as ONE launch graph (Solver._graph_step), the optimizer step behind it; bit-identical to the eager
    loop, also over a second fit with other loss terms. """
    def odevar(f, x):
        return pa.D(f, x) - 2 * np.pi * torch.cos(2 * np.pi * x) + pa.V('new_var', data=torch.Tensor([1.0]))

    def run(graph):
        monkeypatch.setenv('PYDENS_AMD_STEP_GRAPH', '1' if graph else '0')
        monkeypatch.setattr(pa.Solver, 'GENERIC_GRAPH_MIN_REPLAYS', 0)      # (these fits are short on purpose: record whatever is left)
        torch.manual_seed(23)
        solver = pa.Solver(odevar, ndims=1, initial_condition=1, constraints=lambda f, x: f(torch.tensor([0.5])))
        solver.use_fused = False
        solver.fit(niters=15, batch_size=500, lr=0.01, loss_terms=['equation', 'constraint_0'])
        assert solver.last_fit_path == 'generic'
        st = getattr(solver, '_generic_graph', None) or {}
        replays, err = st.get('replays', 0), st.get('error')
        solver.fit(niters=9, batch_size=500, lr=0.01, loss_terms='constraint_0', optimizer=None)
        st = getattr(solver, '_generic_graph', None) or {}
        return (np.array([float(v) for v in solver.losses]), solver.model.flat.detach().cpu().numpy().copy(), replays + st.get('replays', 0),
                err or st.get('error'))
    l0, p0, r0, _ = run(False)
    l1, p1, r1, err = run(True)
    assert np.array_equal(l0, l1) and np.array_equal(p0, p1)
    assert r0 == 0
    # (a constraint that builds its point from host memory inside the step -- torch.tensor([0.5]) -- may be refused by the capture: the
    #  step then stays eager, which the equalities above cover as well; when it is recorded, every iteration behind the warm-up replays)
    assert r1 in (0, (15 - 3) + (9 - 3)), (r1, err)
    print('launch-graph replays:', r1, 'refused:', err)


def test_generic_launch_graph_covers_direction_groups_and_leaves_inner_autograd_eager(pa, monkeypatch):
    """ several kernel calls per half (direction groups: two third-order columns) are recorded and replayed like the plain case,
    bit-identical to the eager loop; steps that differentiate inside their torch code (a callable IC holding a variable, a model
    subclass's own forward()) are never recorded (recording them crashes inside the HIP runtime) and run eagerly. """
    D = pa.D

    def build(which):
        if which == 'direction_groups':
            eq = lambda f, x, y, t: D(f, t) + 0.05 * D(D(D(f, x), x), x) + 0.02 * D(D(D(f, y), y), y) + f * D(f, x)
            return pa.Solver(eq, ndims=3, boundary_condition=0.0, initial_condition=0.2, layout='fa fa f', features=[24, 24, 1], activation='Tanh')
        if which == 'callable_ic':
            ic = lambda x: pa.V('amp', data=torch.Tensor([0.8])) * torch.sin(np.pi * x)
            solver = pa.Solver(lambda f, x, t: D(f, t) - 0.1 * D(D(f, x), x), ndims=2, boundary_condition=0.0, initial_condition=ic,
                               layout='fa fa f', features=[24, 24, 1], activation='Tanh')
            solver.program = None
            solver.use_fused = False
            return solver

        class Scaled(pa.ConvBlockModel):
            def forward(self, xs):
                return self.anzatc(self.conv_block(xs), xs) * (1.0 + 0.5 * xs[:, :1])
        eq = lambda f, x, y: D(D(f, x), x) + D(D(f, y), y) - 5 * torch.sin(np.pi * (x + y))
        return pa.Solver(eq, ndims=2, boundary_condition=1, model=Scaled, layout='fa fa f', features=[24, 24, 1], activation='Tanh')

    def run(which, graph):
        monkeypatch.setenv('PYDENS_AMD_STEP_GRAPH', '1' if graph else '0')
        monkeypatch.setattr(pa.Solver, 'GENERIC_GRAPH_MIN_REPLAYS', 0)      # (these fits are short on purpose: record whatever is left)
        torch.manual_seed(24)
        solver = build(which)
        solver.fit(niters=12, batch_size=600, lr=0.005)
        assert solver.last_fit_path == 'generic'
        st = getattr(solver, '_generic_graph', None) or {}
        return (np.array([float(v) for v in solver.losses]), solver.model.flat.detach().cpu().numpy().copy(), st.get('replays', 0), st.get('error'))
    l0, p0, r0, _ = run('direction_groups', False)
    l1, p1, r1, err = run('direction_groups', True)
    assert np.array_equal(l0, l1) and np.array_equal(p0, p1)
    assert r0 == 0 and r1 == 12 - 3, (r1, err)
    for which in ('callable_ic', 'custom_forward'):
        _, _, replays, err = run(which, True)
        assert replays == 0 and err is None, (which, replays, err)


def test_generic_launch_graph_policy_short_fits_stay_eager_and_replays_are_rechecked(pa, monkeypatch):
    """ round 5: a recording is only made when enough iterations of the fit call are left to replay it -- `for epoch:
    solver.fit(niters=20)` must not pay a capture per call -- and every GENERIC_GRAPH_CHECK_EVERY replays the batch is also stepped
    eagerly and compared bit for bit: an equation whose host-side state changes during the fit (a closure scalar) is caught, the
    fit goes on eagerly and ends where the eager loop ends. """
    monkeypatch.setenv('PYDENS_AMD_STEP_GRAPH', '1')
    torch.manual_seed(25)
    cfg, solver = make_solver('cfg2', pa)
    solver.program = None
    for _ in range(3):
        solver.fit(niters=20, batch_size=1024, lr=0.005)
        st = getattr(solver, '_generic_graph', None) or {}
        assert st.get('graph') is None and st.get('replays', 0) == 0 and not st.get('failed')
    solver.fit(niters=3 + 130, batch_size=1024, lr=0.005)
    st = solver._generic_graph
    assert st['replays'] == 130 and not st['failed'], st.get('error')           # (re-checks at replays 2, 8, 32, 64 and 128 passed on the way)

    # host-side state that moves during the fit: the recorded step keeps the old value, the re-check notices
    state = {'k': 5.0}

    def eq(f, x, y):
        return pa.D(pa.D(f, x), x) + pa.D(pa.D(f, y), y) - state['k'] * torch.sin(np.pi * (x + y))

    class Drift:
        """ sampler that changes the closure scalar half-way through (the reference re-reads it every iteration) """
        def __init__(self):
            self.calls, self.rng = 0, np.random.RandomState(3)

        def sample(self, size):
            self.calls += 1
            if self.calls == 40:
                state['k'] = 7.0
            return self.rng.rand(size, 2)

    def run(graph):
        monkeypatch.setenv('PYDENS_AMD_STEP_GRAPH', '1' if graph else '0')
        state['k'] = 5.0
        torch.manual_seed(26)
        s2 = pa.Solver(eq, ndims=2, boundary_condition=1, layout='fa fa f', features=[20, 20, 1], activation='Tanh')
        s2.program = None
        s2.fit(niters=140, batch_size=512, sampler=Drift(), lr=0.005)
        return np.array([float(v) for v in s2.losses]), (getattr(s2, '_generic_graph', None) or {})
    l_eager, _ = run(False)
    with pytest.warns(RuntimeWarning, match='eager step differ'):
        l_graph, st = run(True)
    assert st['failed'] and st['replays'] == pa.Solver.GENERIC_GRAPH_CHECK_EVERY
    # before the change and from the re-check on the two loops agree bit for bit; in between the replay used the frozen scalar
    assert np.array_equal(l_eager[:39], l_graph[:39])
    assert np.isfinite(l_graph).all() and abs(l_graph[-1] - l_eager[-1]) < 0.5 * abs(l_eager[-1]) + 1.0


@pytest.mark.parametrize('batch', [1, 17, 4097])
def test_launch_graphs_at_tiny_and_boundary_batches(pa, batch, monkeypatch):
    """ a batch below one 16-point tile, one just above it and one just above the fused path's graph threshold: chunk graphs (fused) and
    step graphs (generic) against the eager loops, bit for bit (tools/tiny_batch_graph_check.py is the same check as a script) """
    def run(graph, generic):
        monkeypatch.setenv('PYDENS_AMD_FIT_GRAPH', '1' if graph else '0')
        monkeypatch.setenv('PYDENS_AMD_STEP_GRAPH', '1' if graph else '0')
        monkeypatch.setenv('PYDENS_AMD_FIT_PERSIST', '0')
        monkeypatch.setattr(pa.Solver, 'GENERIC_GRAPH_MIN_REPLAYS', 0)      # (these fits are short on purpose: record whatever is left)
        torch.manual_seed(3)
        solver = pa.Solver(lambda f, x, y: pa.D(pa.D(f, x), x) + pa.D(pa.D(f, y), y) - 5 * torch.sin(np.pi * (x + y)), ndims=2,
                           boundary_condition=1, layout='fa fa f', features=[20, 20, 1], activation='Tanh')
        if generic:
            solver.program = None
        solver.fit(niters=260, batch_size=batch, lr=0.005)
        assert solver.last_fit_path == ('generic' if generic else 'fused')
        return np.array([float(v) for v in solver.losses]), solver.model.flat.detach().cpu().numpy().copy()
    for generic in (False, True):
        l0, p0 = run(False, generic)
        l1, p1 = run(True, generic)
        assert np.isfinite(l1).all()
        assert np.array_equal(l0, l1) and np.array_equal(p0, p1), generic
