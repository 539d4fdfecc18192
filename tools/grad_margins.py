""" Achieved gradient / loss / field errors of the HIP path against the reference-generated goldens (tests/golden), per fixture and
GEMM mode: what the bounds of tests/test_gpu_parity.py are set from (SURVEY 8c item 3 asks 1e-5 rel-L2 per gradient tensor). """
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np      # noqa: E402
import torch            # noqa: E402
import pydens_amd as pa     # noqa: E402
from conftest import Golden, GOLDEN_NAMES, rel_l2   # noqa: E402
from helpers import export_grads, load_params, make_solver   # noqa: E402

for name in GOLDEN_NAMES:
    g = Golden(name)
    for gemm in ('fp32', 'bf16x3'):
        _, solver = make_solver(name, pa)
        solver.set_gemm_mode(gemm)
        load_params(solver, g.params)
        xs = torch.from_numpy(g.points[0].copy()).cuda()
        solver._fused_step(xs, 1)
        lay = solver.model.net.layout
        loss = float(solver.grads[lay.off_loss])
        kernel = solver.model.net.lib.pinn_last_kernel_name().decode()
        if gemm == 'bf16x3' and not int(kernel.rstrip('>').split(',')[-1]) & 512:
            continue                        # no split kernel for this shape: same kernel as fp32
        errs = [rel_l2(got, want) for got, want in zip(export_grads(solver), g.grads) if want is not None]
        norms = [float(np.linalg.norm(want)) for want in g.grads if want is not None]
        print(f'{name:12s} {gemm:7s} loss rel {abs(loss - g.loss0) / g.loss0:.1e}  grad rel-L2 max {max(errs):.1e} '
              f'(tensor norms {min(norms):.1e} .. {max(norms):.1e})  {kernel}', flush=True)
