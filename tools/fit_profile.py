""" Where does an iteration of Solver.fit go at tiny batches (BASELINE config 1, 100 points)? cProfile of the host loop + the rate;
run under `rocprofv3 --kernel-trace --stats` for the device side. usage: python tools/fit_profile.py [niters] """
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch            # noqa: E402
import pinn_configs as pc   # noqa: E402
import pydens_amd as pa     # noqa: E402

niters = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
torch.manual_seed(0)
cfg = pc.make_config('cfg1', pa.D, torch)
solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'])
solver.fit(niters=500, batch_size=100)
torch.cuda.synchronize()
t0 = time.perf_counter()
solver.fit(niters=niters, batch_size=100)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'cfg1 batch 100: {niters / t_all:.0f} it/s ({t_all / niters * 1e6:.2f} us/it; host loop alone {t_host / niters * 1e6:.2f} us/it)')
prof = cProfile.Profile()
prof.enable()
solver.fit(niters=niters, batch_size=100)
prof.disable()
torch.cuda.synchronize()
pstats.Stats(prof).sort_stats('tottime').print_stats(14)
