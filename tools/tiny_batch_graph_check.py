""" Launch graphs (fused chunks, generic steps) against the eager loops at tiny / boundary batch sizes (1, 7, 17, 4096, 4097 points): bit-identical?
The GPU suite runs the same check (tests/test_gpu_parity.py::test_launch_graphs_at_tiny_and_boundary_batches).  usage: python tools/tiny_batch_graph_check.py """
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import pydens_amd as pa
D = pa.D
def run(graph, batch, generic):
    os.environ['PYDENS_AMD_FIT_GRAPH'] = os.environ['PYDENS_AMD_STEP_GRAPH'] = '1' if graph else '0'
    torch.manual_seed(3)
    s = pa.Solver(lambda f, x, y: D(D(f, x), x) + D(D(f, y), y) - 5 * torch.sin(np.pi * (x + y)), ndims=2, boundary_condition=1,
                  layout='fa fa f', features=[20, 20, 1], activation='Tanh')
    if generic:
        s.program = None
    s.fit(niters=300, batch_size=batch, lr=0.005)
    return np.array([float(v) for v in s.losses]), s.model.flat.detach().cpu().numpy().copy(), s.last_fit_path
for batch in (1, 7, 17, 4096, 4097):
    for generic in (False, True):
        l0, p0, path = run(False, batch, generic)
        l1, p1, _ = run(True, batch, generic)
        print(batch, path, 'identical' if np.array_equal(l0, l1) and np.array_equal(p0, p1) else 'DIFFERENT', 'finite' if np.isfinite(l1).all() else 'NAN', flush=True)
