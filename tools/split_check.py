""" split-bf16 kernels against the exact-fp32 kernels on the device: same points, same parameters, loss and every gradient tensor
(tools: python tools/split_check.py [lib.so]); the parity tests proper are tests/test_gpu_parity.py """
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np      # noqa: E402
import torch            # noqa: E402
import pinn_configs as pc   # noqa: E402
import pydens_amd as pa     # noqa: E402
from pydens_amd import engine   # noqa: E402
from helpers import export_grads   # noqa: E402

lib = engine.bind(ctypes.CDLL(os.path.abspath(sys.argv[1]))) if len(sys.argv) > 1 else engine.load_library()
for name, n in (('cfg2', 65536), ('cfg4', 131072), ('cfg2', 1000), ('cfg4', 77)):
    torch.manual_seed(0)
    cfg = pc.make_config(name, pa.D, torch)
    solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], _lib=lib)
    pts = torch.from_numpy(pc.sample_points(cfg, n, seed=3)).cuda()
    res = {}
    for mode in ('fp32', 'bf16x3'):
        solver.model.net.set_gemm_mode(mode)
        solver._fused_step(pts, 1)
        torch.cuda.synchronize()
        lay = solver.model.net.layout
        res[mode] = (float(solver.grads[lay.off_loss]), [g.astype(np.float64) for g in export_grads(solver)],
                     lib.pinn_last_kernel_name().decode())
    rel = [float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)) for a, b in zip(res['bf16x3'][1], res['fp32'][1])]
    print(f"{name} n={n}: {res['bf16x3'][2]} vs {res['fp32'][2]}: loss {res['bf16x3'][0]:.8g} vs {res['fp32'][0]:.8g}, "
          f"max gradient rel-L2 {max(rel):.2e}", flush=True)
