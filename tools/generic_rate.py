""" Solver.fit rate of the tutorial's variable + constraint problem (cells 50-60): equation only and with the constraint
term, on the fused path (residual programs; default) and on the generic path (use_fused = False: kernel streams + the
user's torch code + autograd). """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pydens_amd as pa
from pydens_amd import D, V

def odevar(f, x):
    return D(f, x) - 2 * np.pi * torch.cos(2 * np.pi * x) + V('new_var', data=torch.Tensor([1.0]))

# (PYDENS_AMD_STEP_GRAPH=0: the eager loops; default: the gradient part of an iteration replayed as one launch graph, round 4)
for batch in (500, 65536):
    solver = pa.Solver(odevar, ndims=1, initial_condition=1, constraints=lambda f, x: f(torch.tensor([0.5])))
    for terms, fused in (('equation', True), (['equation', 'constraint_0'], True), (['equation', 'constraint_0'], False)):
        solver.use_fused = fused
        solver.fit(niters=20, batch_size=batch, lr=0.01, loss_terms=terms)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.fit(niters=200, batch_size=batch, lr=0.01, loss_terms=terms)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
        st = getattr(solver, '_generic_graph', None) or {}
        print(f'batch {batch:6d} terms {terms!s:32s} path {solver.last_fit_path:8s} {dt * 1e3:8.3f} ms/it   '
              f'({st.get("replays", 0)} launch-graph replays{", refused: " + st["error"] if st.get("error") else ""})', flush=True)
