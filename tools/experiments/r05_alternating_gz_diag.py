""" cfg3 on gpurun_variants/lib_altgz.so differs from the product (round 4, last GPU minutes): run-to-run (a race) or consistently (logic)? """
import ctypes, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import pinn_configs as pc
import pydens_amd as pa
from pydens_amd import engine
libs = {'product': engine.load_library(), 'altgz': engine.bind(ctypes.CDLL('/root/repo/gpurun_variants/lib_altgz.so'))}
for name, n in (('cfg3', 262144), ('cfg3', 4096), ('cfg3', 16)):
    out = {}
    for tag, lib in libs.items():
        torch.manual_seed(0)
        cfg = pc.make_config(name, pa.D, torch)
        s = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], lib=lib)
        xs = torch.from_numpy(pc.sample_points(cfg, n, seed=1)).cuda()
        runs = []
        for _ in range(6):
            s._fused_step(xs, 1); runs.append(s.grads.clone())
        out[tag] = runs
    ref = out['product'][0]
    print(name, n, 'product repeatable:', all(torch.equal(r, ref) for r in out['product']),
          '| altgz repeatable:', all(torch.equal(r, out['altgz'][0]) for r in out['altgz']),
          '| max |altgz - product| per run:', ['%.2e' % float((r - ref).abs().max()) for r in out['altgz']],
          '| |g|max %.2e' % float(ref.abs().max()), flush=True)
