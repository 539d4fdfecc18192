""" Solver.fit rate of ONE BASELINE config (python tools/fit_one.py cfg3 [iters]); rocprofv3-friendly. """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pinn_configs as pc
import pydens_amd as pa

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = {'cfg1': 100, 'cfg2': 65536, 'cfg3': 262144, 'cfg4': 131072, 'cfg5': 131072}[name]
torch.manual_seed(0)
cfg = pc.make_config(name, pa.D, torch)
solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'])
sampler = pa.NumpySampler('uniform') & pa.NumpySampler('uniform', low=1, high=5) if name == 'cfg4' else None
import gc
gc.collect()            # (in front of the warm-up: the collection itself leaves the GPU idle for 40 ms, which costs the clocks)
gc.freeze()
gc.disable()
solver.fit(niters=20, batch_size=n, sampler=sampler)
torch.cuda.synchronize()
# (as `timeit` does: a full collection of Python's cyclic garbage collector stops the launching thread for ~40 ms in a process with torch loaded;
#  whether one falls into the 64 ms this call of BASELINE config 4 takes depended on the allocation count of the host code -- round 6)
t0 = time.perf_counter()
solver.fit(niters=iters, batch_size=n, sampler=sampler)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
gc.enable()
losses = solver.losses
print(f'{name}: Solver.fit {iters / dt:9.1f} it/s  {n * iters / dt:12.4g} points/s  ({dt / iters * 1e3:.3f} ms/it, batch {n}, path {solver.last_fit_path}, '
      f'loss {float(losses[20]):.4g} -> {float(losses[-1]):.4g})')
import ctypes
st = (ctypes.c_int32 * 4)()
solver.model.net.lib.pinn_debug_fit_graph_stats(st)
print(f'{name}: launch graphs: {st[0]} chunks replayed, {st[1]} captured, {st[2]} refused (HIP error {st[3]})')
