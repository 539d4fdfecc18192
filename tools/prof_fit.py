""" Host-side profile (cProfile) of a small-batch Solver.fit (BASELINE cfg1, 100 points): where the Python time of the
latency-bound regime goes. """
import sys, time, cProfile, pstats
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pinn_configs as pc, pydens_amd as pa
torch.manual_seed(0)
cfg = pc.make_config('cfg1', pa.D, torch)
solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'])
solver.fit(niters=200, batch_size=100)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
solver.fit(niters=3000, batch_size=100)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
t0 = time.perf_counter(); solver.fit(niters=3000, batch_size=100); torch.cuda.synchronize(); print('us/it', (time.perf_counter() - t0) / 3000 * 1e6)
