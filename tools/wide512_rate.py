""" step time of hidden width 512 (round 6): a 1-D second-order ODE (fused path, one kernel call of S = 3 streams) and the 2-D Poisson problem
(generic path, one call per second-order direction) on a 4 x 512 Tanh net at 65 536 points; executed fp32 MFMA fraction of 157.3 TF """
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
import pydens_amd as pa     # noqa: E402
from pydens_amd import D    # noqa: E402

N = 65536
PEAK = 157.3


def run(name, eq, kw, path, streams_exec):
    torch.manual_seed(0)
    solver = pa.Solver(eq, **kw)
    d = kw['ndims']
    xs = torch.rand((N, d), device='cuda')
    mse = torch.nn.MSELoss()
    step = (lambda: solver._fused_step(xs, 1)) if path == 'fused' else (lambda: solver._generic_step(xs, ('equation',), [], mse, 1))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    dims = solver.model.layer_dims
    hidden = sum(a * b for a, b in zip(dims[1:-2], dims[2:-1]))
    flops = 6.0 * streams_exec * hidden * N
    lib = solver.model.net.lib
    print(f'{name:16s} {ms:8.3f} ms / step   {flops / (ms * 1e-3) / 1e12:6.1f} TF executed = {flops / (ms * 1e-3) / 1e12 / PEAK:.3f} of {PEAK} TF   '
          f'{lib.pinn_last_kernel_name().decode()}  {lib.pinn_last_wgrad_kernel_name().decode()}', flush=True)


if __name__ == '__main__':
    net = dict(layout='fa fa fa fa f', features=[512, 512, 512, 512, 1], activation='Tanh')
    run('ode_fused_512', lambda f, x: D(D(f, x), x) + f - torch.sin(3.0 * x), dict(ndims=1, boundary_condition=0.2, **net), 'fused', 3)
    run('poisson_512', lambda f, x, y: D(D(f, x), x) + D(D(f, y), y) - 5 * torch.sin(np.pi * (x + y)), dict(ndims=2, boundary_condition=1, **net),
        'generic', 6)       # (two calls of S = 3 streams each: the value stream rides twice)
    net256 = dict(layout='fa fa fa fa f', features=[256, 256, 256, 256, 1], activation='Tanh')
    run('ode_fused_256', lambda f, x: D(D(f, x), x) + f - torch.sin(3.0 * x), dict(ndims=1, boundary_condition=0.2, **net256), 'fused', 3)
