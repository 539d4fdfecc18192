""" Data-parallel path: one process per device, each rank steps on its shard, ONE all-reduce of the flat
gradient buffer per iteration. Covered on CPU with gloo through the emulated kernels at world sizes 1, 2, 4 and 8 (SURVEY section 4
iv: 1 / 2 / 8): R ranks on their shares of every batch must follow the single-process trajectory on the full batches (same
mean-square loss, same Adam), also when the shares are uneven (1 003 points over 4 / 8 ranks, 99 over 2). The communicator's
bounded bootstrap (ncclCommInitRank that never returns on one rank -> every rank falls back together) is driven with a stand-in
library on a gloo group. """
import os
import socket
import sys
import tempfile

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import Golden, rel_l2

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, path):
    import ctypes
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(HERE, 'emu'))
    import torch.distributed as dist
    import build_emu
    import pydens_amd as pa
    from pydens_amd import engine
    from helpers import FixedBatches, export_params, load_params, make_solver
    torch.set_num_threads(1)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lib = engine.bind(ctypes.CDLL(build_emu.build()))
    solver, points, fit_kw, start = _problem(path, pa, lib)
    if rank == 0:
        load_params(solver, start)                  # the other rank starts elsewhere: fit must broadcast from rank 0
    shard = points[:, rank::world]                  # rank r owns points r, r+world, ... of every batch
    # `batch_size` keeps the reference's meaning under data parallelism: the GLOBAL points per iteration
    solver.fit(niters=points.shape[0], batch_size=points.shape[1], sampler=FixedBatches(shard), **fit_kw)
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), losses=np.array([float(v) for v in solver.losses]),
             variables=_variables(solver), **{f'p{i}': p for i, p in enumerate(export_params(solver))})
    dist.destroy_process_group()


def _variables(solver):
    return np.array([float(getattr(solver.model, name).detach()) for name in sorted(solver.model.variables)])


def _problem(path, pa, lib):
    """ -> (solver, batches [steps, N, d], fit kwargs, start parameters) """
    from helpers import make_solver
    if path in ('fused', 'generic', 'fused_uneven', 'generic_uneven', 'fused_1003', 'generic_1003'):
        g = Golden('cfg1')
        _, solver = make_solver('cfg1', pa, _lib=lib, device='cpu')
        if path.startswith('generic'):
            solver.program = None
        # `_uneven`: 99 points per iteration on two ranks -- shares of 50 and 49 points, weighted by the GLOBAL count
        points = g.points[:3, :99] if path.endswith('_uneven') else g.points[:3]
        if path.endswith('_1003'):
            # 1 003 points per iteration: 4 ranks own 251 / 251 / 251 / 250 of them, 8 ranks 126 x 3 / 125 x 5 (ragged last tiles everywhere)
            points = np.random.RandomState(1003).rand(2, 1003, 2).astype(np.float32)
        return solver, np.ascontiguousarray(points), dict(lr=g.lr), g.params
    if path == 'custom_forward_generic':
        # the model plug-in seam with a forward() of its own (model_torch.py:52-54): bare network on the kernels, the ansatz (with
        # its log_scale gradient from autograd) and an output factor as torch code, generic step path -- under data parallelism
        class Scaled(pa.ConvBlockModel):
            def forward(self, xs):
                return self.anzatc(self.conv_block(2.0 * xs - 1.0), xs) * (1.0 + 0.5 * xs[:, :1])      # (inputs normalised in front of the net: Solver._input_map)
        torch.manual_seed(5)
        eq = lambda f, x, t: pa.D(f, t) - 0.1 * pa.D(pa.D(f, x), x) + f * f
        solver = pa.Solver(eq, ndims=2, boundary_condition=0.2, initial_condition=lambda x: torch.sin(np.pi * x), model=Scaled,
                           layout='fa fa f', features=[12, 10, 1], activation='Tanh', _lib=lib, device='cpu')
        rng = np.random.RandomState(6)
        start = [np.asarray(rng.randn(*p.shape) * 0.5, dtype=np.float32) for p in export_params_of(solver)]
        return solver, rng.rand(3, 32, 2).astype(np.float32), dict(lr=0.02), start
    # tutorial cells 50-60: trainable variable in the equation + a constraint term; every rank evaluates the constraint,
    # the all-reduce sums the copies, hence its 1 / world scale
    from test_emu_engine import _variable_problem
    torch.manual_seed(21)
    eq, con = _variable_problem(pa.D, pa.V, torch)
    solver = pa.Solver(eq, constraints=con, ndims=1, initial_condition=1, layout='fafaf', features=[12, 10, 1],
                       activation='Tanh', _lib=lib, device='cpu')
    solver.use_fused = path == 'constraint_fused'
    rng = np.random.RandomState(3)
    start = [np.asarray(rng.randn(*p.shape) * 0.5, dtype=np.float32) for p in export_params_of(solver)]
    return solver, rng.rand(4, 32, 1).astype(np.float32), dict(lr=0.05, loss_terms=['equation', 'constraint_0']), start


def export_params_of(solver):
    from helpers import export_params
    return export_params(solver)


def _run(path, world=2):
    import ctypes
    sys.path.insert(0, os.path.join(HERE, 'emu'))
    import build_emu
    import pydens_amd as pa
    from pydens_amd import engine
    from helpers import FixedBatches, export_params, load_params
    lib = engine.bind(ctypes.CDLL(build_emu.build()))
    single, points, fit_kw, start = _problem(path, pa, lib)
    load_params(single, start)
    single.fit(niters=points.shape[0], batch_size=points.shape[1], sampler=FixedBatches(points), **fit_kw)
    assert single.last_fit_path == ('generic' if 'generic' in path else 'fused')
    want_losses = np.array([float(v) for v in single.losses])
    want, want_vars = export_params(single), _variables(single)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, _free_port(), tmp, path), nprocs=world, join=True)
        for rank in range(world):
            z = np.load(os.path.join(tmp, f'rank{rank}.npz'))
            np.testing.assert_allclose(z['losses'], want_losses, rtol=1e-5)      # loss slot is all-reduced too
            np.testing.assert_allclose(z['variables'], want_vars, rtol=1e-5, atol=1e-6)
            for i, w in enumerate(want):
                assert rel_l2(z[f'p{i}'], w) < 1e-5, (rank, i)


def test_two_ranks_fused_path_equals_single_process():
    _run('fused')


def test_two_ranks_generic_path_equals_single_process():
    _run('generic')


def test_two_ranks_uneven_shares_of_the_global_batch():
    _run('fused_uneven')
    _run('generic_uneven')


def test_two_ranks_constraint_term_and_variable_fused():
    _run('constraint_fused')


def test_two_ranks_constraint_term_and_variable_generic():
    _run('constraint_generic')


def test_two_ranks_model_with_its_own_forward():
    _run('custom_forward_generic')


def test_one_rank_group_equals_single_process():
    _run('fused', world=1)
    _run('generic', world=1)


def test_four_ranks_uneven_shares_of_1003_points():
    _run('fused_1003', world=4)
    _run('generic_1003', world=4)


def test_eight_ranks_uneven_shares_of_1003_points():
    _run('fused_1003', world=8)
    _run('generic_1003', world=8)


def test_eight_ranks_constraint_term_and_variable():
    _run('constraint_fused', world=8)


# ---- the communicator's bounded bootstrap -----------------------------------------------------------------------------------------
class _FakeRccl:
    """ stand-in for librccl with the calling convention comm.py uses: rank `stuck` never returns from ncclCommInitRank in time """
    def __init__(self, rank, stuck, log):
        self.rank, self.stuck, self.log = rank, stuck, log

    def ncclGetUniqueId(self, uid_ref):
        uid = uid_ref._obj
        for i in range(128):
            uid.internal[i] = (i * 7 + 1) & 255
        return 0

    def ncclCommInitRank(self, comm_ref, world, uid, rank):
        import time
        assert bytes(uid.internal) == bytes(((i * 7 + 1) & 255) for i in range(128))          # the id arrived whole on every rank
        if rank in self.stuck:
            time.sleep(3.0)                    # (longer than PYDENS_AMD_COMM_TIMEOUT below, short enough for the late-abort check)
        comm_ref._obj.value = 4242 + rank
        return 0

    def ncclCommAbort(self, comm):
        self.log.append(('abort', comm.value))
        return 0

    def ncclCommDestroy(self, comm):
        self.log.append(('destroy', comm.value if hasattr(comm, 'value') else comm))
        return 0

    def ncclGetErrorString(self, rc):
        return b'fake'


def _comm_worker(rank, world, port, out_dir, stuck):
    import json
    import time
    import warnings
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    from pydens_amd import comm
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), PYDENS_AMD_COMM_TIMEOUT='1')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    log = []
    fake = _FakeRccl(rank, stuck, log)
    comm._rccl = lambda: fake
    comm.Communicator._FORCE_DIRECT = True
    t0 = time.time()
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter('always')
        c = comm.Communicator(torch.device('cpu'))
    took = time.time() - t0
    x = torch.full((5,), float(rank + 1))
    c.all_reduce_(x)                            # the agreed fallback path works on every rank
    worker, outcome = c._late_init
    worker.join(10.0)                           # the abandoned call returns later: its thread must abort what it got
    json.dump({'direct': c.direct, 'reason': c.fallback_reason, 'took': took, 'sum': x.tolist(), 'log': log,
               'warned': [str(w.message) for w in seen], 'late': bool(outcome.get('aborted_late')), 'describe': c.describe()},
              open(os.path.join(out_dir, f'comm{rank}.json'), 'w'))
    dist.destroy_process_group()


def test_communicator_bootstrap_timeout_makes_every_rank_fall_back_together():
    import json
    world, stuck = 4, (2,)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_comm_worker, args=(world, _free_port(), tmp, stuck), nprocs=world, join=True)
        for rank in range(world):
            r = json.load(open(os.path.join(tmp, f'comm{rank}.json')))
            assert r['direct'] is False and 'timed out' in r['reason'], r
            assert r['took'] < 2.5, r['took']                       # bounded by PYDENS_AMD_COMM_TIMEOUT, not by the stuck call
            assert r['sum'] == [float(sum(range(1, world + 1)))] * 5
            assert any('torch.distributed' in w for w in r['warned'])
            assert 'fallback' in r['describe']['all_reduce']
            if rank in stuck:
                assert r['late'] and ('abort', 4242 + rank) in [tuple(e) for e in r['log']]      # no leaked communicator
            else:
                assert ('destroy', 4242 + rank) in [tuple(e) for e in r['log']]                  # made, agreed "failed", destroyed
