""" workload for rocprofv3 passes: a handful of fused residual steps (+ Adam) of one BASELINE config. """
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                    # noqa: E402
import pinn_configs as pc       # noqa: E402
import pydens_amd as pa         # noqa: E402
from pydens_amd.solver import FlatAdam  # noqa: E402

cfg_name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(0)
cfg = pc.make_config(cfg_name, pa.D, torch)
solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'])
n = min(cfg['n_points'], 131072)
xs = torch.from_numpy(pc.sample_points(cfg, n, seed=1)).cuda()
solver.optimizer = FlatAdam(solver.model, lr=0.005)
solver.optimizer.refresh()
for _ in range(steps):
    solver._fused_step(xs, 1, adam=solver.optimizer)
torch.cuda.synchronize()
print('done', float(solver.grads[solver.model.net.layout.off_loss]))
