""" Data-parallel path: one process per device, each rank steps on its shard, ONE all-reduce of the flat
gradient buffer per iteration. Covered on CPU with gloo, world_size 2, through the emulated kernels: two ranks on
half batches must follow the single-process trajectory on the full batches (same mean-square loss, same Adam). """
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
    g = Golden('cfg1')
    _, solver = make_solver('cfg1', pa, lib=lib, device='cpu')
    if rank == 0:
        load_params(solver, g.params)               # the other rank starts elsewhere: fit must broadcast from rank 0
    shard = g.points[:3, rank::world]               # rank r owns points r, r+world, ... of every batch
    if path == 'generic':
        solver.program = None
    solver.fit(niters=3, batch_size=shard.shape[1], sampler=FixedBatches(shard), lr=g.lr)
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), losses=np.array([float(v) for v in solver.losses]),
             **{f'p{i}': p for i, p in enumerate(export_params(solver))})
    dist.destroy_process_group()


def _run(path):
    import ctypes
    sys.path.insert(0, os.path.join(HERE, 'emu'))
    import build_emu
    import pydens_amd as pa
    from pydens_amd import engine
    from helpers import FixedBatches, export_params, load_params, make_solver
    lib = engine.bind(ctypes.CDLL(build_emu.build()))
    g = Golden('cfg1')
    _, single = make_solver('cfg1', pa, lib=lib, device='cpu')
    load_params(single, g.params)
    if path == 'generic':
        single.program = None
    single.fit(niters=3, batch_size=g.points.shape[1], sampler=FixedBatches(g.points[:3]), lr=g.lr)
    want_losses = np.array([float(v) for v in single.losses])
    want = export_params(single)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, _free_port(), tmp, path), nprocs=2, join=True)
        for rank in range(2):
            z = np.load(os.path.join(tmp, f'rank{rank}.npz'))
            np.testing.assert_allclose(z['losses'], want_losses, rtol=1e-5)      # loss slot is all-reduced too
            for i, w in enumerate(want):
                assert rel_l2(z[f'p{i}'], w) < 1e-5, (rank, i)


def test_two_ranks_fused_path_equals_single_process():
    _run('fused')


def test_two_ranks_generic_path_equals_single_process():
    _run('generic')
