""" Solver.fit rate on the GENERIC step path (pinn_jet_forward -> the user's torch code + autograd -> pinn_jet_backward -> Adam) at the
batch sizes the reference's tutorials use, eager (PYDENS_AMD_STEP_GRAPH=0) against the launch-graph replay (Solver._generic_step_auto).
One fresh process per cell.  usage: python tools/generic_graph_rate.py            (children: ... <graph> <problem> <batch>) """
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) == 1:
    for problem in ('poisson_4x64', 'burgers_vector_V'):
        for batch in (100, 1000, 8192):
            for graph in ('0', '1'):
                out = subprocess.run([sys.executable, os.path.abspath(__file__), graph, problem, str(batch)], capture_output=True, text=True,
                                     env=dict(os.environ, PYDENS_AMD_STEP_GRAPH=graph))
                print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else f'FAILED {out.stderr[-300:]}', flush=True)
    sys.exit(0)

import numpy as np, torch          # noqa: E402
import pydens_amd as pa            # noqa: E402
graph, problem, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
torch.manual_seed(0)
if problem == 'poisson_4x64':
    solver = pa.Solver(lambda f, x, y: pa.D(pa.D(f, x), x) + pa.D(pa.D(f, y), y) - 5 * torch.sin(np.pi * (x + y)), ndims=2, boundary_condition=1,
                       layout='fa fa fa fa f', features=[64, 64, 64, 64, 1], activation='Tanh')
    solver.program = None           # (forced: the tracer would lower this one)
else:
    def eq(f, x, t):
        w = pa.V('w', data=torch.Tensor([0.5, 1.5]))
        return pa.D(f, t) - 0.1 * w[0] * pa.D(pa.D(f, x), x) + w[1] * f * pa.D(f, x)
    solver = pa.Solver(eq, ndims=2, boundary_condition=0.0, initial_condition=0.3, layout='fa fa f', features=[32, 32, 1], activation='Tanh')
iters = 2000 if batch <= 1000 else 600
solver.fit(niters=30, batch_size=batch, lr=0.005)
torch.cuda.synchronize()
import gc; gc.collect(); gc.disable()          # (as timeit does: tools/fit_one.py)
t0 = time.perf_counter()
solver.fit(niters=iters, batch_size=batch, lr=0.005, optimizer=None)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
st = getattr(solver, '_generic_graph', None) or {}
print(f'{problem:18s} batch {batch:5d} graph={graph} path {solver.last_fit_path:8s} {dt * 1e3:7.3f} ms/it  {1 / dt:8.1f} it/s  '
      f'({st.get("replays", 0)} replays{", capture refused: " + st["error"] if st.get("error") else ""})')
