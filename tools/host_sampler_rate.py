""" Solver.fit rate when the collocation batch comes from a HOST sampler (the reference's plug-in: `sampler.sample(n)` returns
a float64 ndarray, model_torch.py:433) -- i.e. with the float32 cast and the PCIe copy of the batch inside every iteration --
next to the on-device Philox sampler. usage: python tools/host_sampler_rate.py [cfg2|cfg4] """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pinn_configs as pc
import pydens_amd as pa

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
n = {'cfg2': 65536, 'cfg4': 131072}[name]
cfg = pc.make_config(name, pa.D, torch)


class HostUniform:                       # no columns(): the solver cannot draw it on the device
    def __init__(self, lo, hi):
        self.lo, self.hi, self.rng = np.asarray(lo, float), np.asarray(hi, float), np.random.RandomState(0)

    def sample(self, size):
        return self.lo + (self.hi - self.lo) * self.rng.rand(size, len(self.lo))


for label, sampler in (('device Philox sampler', None if name == 'cfg2' else pa.NS('u') & pa.NS('u', low=1, high=5)),
                       ('host numpy sampler + PCIe copy', HostUniform(cfg['low'], cfg['high']))):
    torch.manual_seed(0)
    solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'])
    solver.fit(niters=20, batch_size=n, sampler=sampler)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solver.fit(niters=200, batch_size=n, sampler=sampler)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f'{name} {label:32s} {dt * 1e3:.4f} ms/it  {n / dt:.4g} points/s')
