""" Kernel micro-benchmark: times the fused tile kernel (pinn_residual_step) of one or more builds of the library
on a BASELINE config with HIP events (pinn_profile_tile). Usage: python tools/kbench.py [cfg] lib1.so [lib2.so ...] """
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
import pinn_configs as pc   # noqa: E402
import pydens_amd as pa     # noqa: E402
from pydens_amd import engine   # noqa: E402


def bench(cfg_name, lib_path, n=None, reps=30, rounds=3):
    lib = engine.bind(ctypes.CDLL(lib_path))
    torch.manual_seed(0)
    cfg = pc.make_config(cfg_name, pa.D, torch)
    solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], lib=lib)
    n = n or min(cfg['n_points'], 131072)
    xs = torch.from_numpy(pc.sample_points(cfg, n, seed=1)).cuda()
    for _ in range(5):
        solver._fused_step(xs, 1)
    torch.cuda.synchronize()
    lay = solver.model.net.layout
    out = []
    lib.pinn_profile_tile(1)
    for _ in range(rounds):
        ts = []
        for _ in range(reps):
            solver._fused_step(xs, 1)
            ts.append(float(lib.pinn_last_tile_ms()))
        out.append(float(np.median(ts)))
    lib.pinn_profile_tile(0)
    loss = float(solver.grads[lay.off_loss])
    gsum = float(solver.grads[:lay.p_core].double().abs().sum())
    return out, loss, gsum, n


if __name__ == '__main__':
    args = sys.argv[1:]
    cfg_name = 'cfg2'
    if args and not args[0].endswith('.so'):
        cfg_name = args.pop(0)
    libs = args or [engine.library_path()]
    for _ in range(2):                      # interleaved rounds
        for path in libs:
            ms, loss, gsum, n = bench(cfg_name, path)
            print(f'{os.path.basename(path):40s} {cfg_name} n={n} tile ms (median per round) {ms}  loss {loss:.6f} |g|1 {gsum:.6f}',
                  flush=True)
