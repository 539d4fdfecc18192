""" User-facing Solver.fit rate (iterations/s, points/s) for the BASELINE configs, default on-device sampler. """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pinn_configs as pc
import pydens_amd as pa

for name, n, iters in (('cfg1', 100, 2000), ('cfg2', 65536, 300), ('cfg4', 131072, 300), ('cfg3', 262144, 30), ('cfg5', 131072, 20)):
    torch.manual_seed(0)
    cfg = pc.make_config(name, pa.D, torch)
    solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'])
    sampler = None
    if name == 'cfg4':
        sampler = pa.NumpySampler('uniform') & pa.NumpySampler('uniform', low=1, high=5)
    solver.fit(niters=20, batch_size=n, sampler=sampler)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solver.fit(niters=iters, batch_size=n, sampler=sampler)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses = solver.losses
    print(f'{name}: Solver.fit {iters / dt:9.1f} it/s  {n * iters / dt:12.4g} points/s  ({dt / iters * 1e3:.3f} ms/it, path {solver.last_fit_path}, '
          f'loss {float(losses[20]):.4g} -> {float(losses[-1]):.4g})', flush=True)
