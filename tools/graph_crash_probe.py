""" Which part of a direction-group generic step makes hipStreamEndCapture / graph instantiation crash?  Children record growing prefixes of the step.
usage: python tools/graph_crash_probe.py            (children: ... <stage>) """
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    for stage in ('fwd', 'fwd_eq', 'fwd_eq_bwd', 'select', 'select_zero', 'zero_row_only', 'full'):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), stage], capture_output=True, text=True)
        print(f'{stage:14s} rc {out.returncode:4d}  {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr.strip().splitlines()[-1][:160] if out.stderr.strip() else ""}', flush=True)
    sys.exit(0)
import numpy as np, torch          # noqa: E402
import pydens_amd as pa            # noqa: E402
D = pa.D
stage = sys.argv[1]
torch.manual_seed(0)
eq = lambda f, x, y, t: D(f, t) + 0.05 * D(D(D(f, x), x), x) + 0.02 * D(D(D(f, y), y), y) + f * D(f, x)
solver = pa.Solver(eq, ndims=3, boundary_condition=0.0, initial_condition=lambda x, y: torch.sin(np.pi * x) * y * (1 - y),
                   layout='fa fa f', features=[24, 24, 1], activation='Tanh')
model, spec = solver.model, solver.spec
xs = torch.rand(600, 3, device='cuda')
mse = torch.nn.MSELoss()
for _ in range(3):
    solver._generic_step(xs, ('equation',), [], mse, 1)
torch.cuda.synchronize()

def body():
    if stage == 'zero_row_only':
        t = torch.ones(6, 600, device='cuda')
        t[0].zero_()
        return
    if stage == 'full':
        solver._generic_step(xs, ('equation',), [], mse, 1)
        return
    leaf = torch.empty((spec.n_streams, xs.shape[0]), dtype=torch.float32, device='cuda')
    for num, (dirs_g, n2g, idx) in enumerate(spec.groups):
        part = model.net.jet_forward(model.flat, xs, dirs_g, n2g, ic_const=model.kernel_ic_const())
        leaf.index_copy_(0, solver._group_rows(num, idx), part)
    if stage == 'fwd':
        return
    leaf.requires_grad_()
    ic = solver._ic_streams(xs, create_graph=True)
    r = solver._eval_equation(leaf, xs, ic)
    loss = mse(r, torch.zeros_like(xs[:, :1]))
    if stage == 'fwd_eq':
        return
    loss.backward()
    if stage == 'fwd_eq_bwd':
        return
    for num, (dirs_g, n2g, idx) in enumerate(spec.groups):
        gin = leaf.grad.index_select(0, solver._group_rows(num, idx))
        if stage == 'select_zero' and num > 0:
            gin[0].zero_()

g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
g.replay()
torch.cuda.synchronize()
print('recorded and replayed')
