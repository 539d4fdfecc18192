""" Planning probe: does replaying the step as a hipGraph shrink the dependent-launch gaps?  Captures POOL fused steps
(tile kernel + reduction/Adam each, on different resident batches) into one graph and compares the replay rate with the
plain stream launches of bench.py. The Adam step count is baked into the captured launches (every replay repeats steps
1..POOL), so this measures TIME only -- a device-side step counter is what a graph-replayed fit would need. """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pinn_configs as pc
import pydens_amd as pa
from pydens_amd.solver import FlatAdam

POOL, N, REPS = 8, 65536, 25
torch.manual_seed(0)
cfg = pc.make_config('cfg2', pa.D, torch)
solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'])
pool = [torch.rand((N, 2), device='cuda') for _ in range(POOL)]
solver.optimizer = FlatAdam(solver.model, lr=0.005)
solver.optimizer.refresh()


def steps():
    for xs in pool:
        solver._fused_step(xs, 1, adam=solver.optimizer)


for _ in range(3):
    steps()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(REPS):
    steps()
torch.cuda.synchronize()
plain = (time.perf_counter() - t0) / (REPS * POOL)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    steps()
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    steps()
for _ in range(3):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(REPS):
    graph.replay()
torch.cuda.synchronize()
replay = (time.perf_counter() - t0) / (REPS * POOL)
print(f'stream launches {plain * 1e3:.4f} ms/step   hipGraph replay {replay * 1e3:.4f} ms/step   ({(plain / replay - 1) * 100:+.1f} %)')
