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


def bench(cfg_name, lib_path, n=None, reps=None, rounds=3):
    reps = reps or (8 if cfg_name in ('cfg3', 'cfg5') else 30)
    lib = engine.bind(ctypes.CDLL(lib_path))
    torch.manual_seed(0)
    cfg = pc.make_config(cfg_name, pa.D, torch, V=pa.V)
    solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], _lib=lib)
    n = n or (cfg['n_points'] if cfg_name == 'cfg3' else min(cfg['n_points'], 131072))
    xs = torch.from_numpy(pc.sample_points(cfg, n, seed=1)).cuda()
    for _ in range(300 if cfg_name in ('cfg2', 'cfg4') else 10):     # (also brings a fresh process's GPU to its clocks)
        solver._fused_step(xs, 1)
    torch.cuda.synchronize()
    lay = solver.model.net.layout
    out = []
    if FLAGS:
        lib.pinn_debug_set_flags(FLAGS)            # (-DPINN_DEBUG_ABI builds: tools/variant.sh)
    lib.pinn_profile_tile(1)
    wg = []
    for _ in range(rounds):
        ts, tw = [], []
        for _ in range(reps):
            solver._fused_step(xs, 1)
            ts.append(float(lib.pinn_last_tile_ms()))
            tw.append(float(lib.pinn_last_wgrad_ms()))
        out.append(round(float(np.median(ts)), 4))
        wg.append(round(max(float(np.median(tw)), 0.0), 4))
    if max(wg) > 0:
        out = [f'{a:.4f}+{b:.4f}={a + b:.4f}' for a, b in zip(out, wg)]
    lib.pinn_profile_tile(0)
    if FLAGS:
        lib.pinn_debug_set_flags(0)
    loss = float(solver.grads[lay.off_loss])
    gsum = float(solver.grads[:lay.p_core].double().abs().sum())
    return out, loss, gsum, n


FLAGS = 0


if __name__ == '__main__':
    args = sys.argv[1:]
    if args and args[0].startswith('--flags='):
        FLAGS = int(args.pop(0).split('=')[1])
    cfg_name = 'cfg2'
    if args and not args[0].endswith('.so'):
        cfg_name = args.pop(0)
    libs = args or [engine.library_path()]
    for _ in range(2):                      # interleaved rounds
        for path in libs:
            ms, loss, gsum, n = bench(cfg_name, path)
            print(f'{os.path.basename(path):40s} {cfg_name} n={n} tile ms (median per round) {ms}  loss {loss:.6f} |g|1 {gsum:.6f}',
                  flush=True)
