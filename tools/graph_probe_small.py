""" Planning probe for the latency regime (BASELINE config 1, batch 100): would replaying a chunk of fit iterations (sample -> tile ->
reduce + Adam, K times: pinn_fit_steps) as ONE hipGraph shrink the dependent-launch gaps? The per-iteration arguments (batch counter,
Adam step, loss slot) are baked into the captured launches, so a replay repeats the same K iterations: this measures TIME only. """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pinn_configs as pc
import pydens_amd as pa
from pydens_amd.solver import FlatAdam

name, batch, K, REPS = (sys.argv[1] if len(sys.argv) > 1 else 'cfg1'), int(sys.argv[2]) if len(sys.argv) > 2 else 100, 128, 20
torch.manual_seed(0)
cfg = pc.make_config(name, pa.D, torch)
solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'])
solver.fit(niters=4, batch_size=batch, lr=0.005)             # builds optimizer, sampler columns, workspace
model, spec, adam = solver.model, solver.spec, solver.optimizer
columns = solver._device_columns(None)
comb_w = solver.residual_plan.comb_w if solver.residual_plan is not None else None
n2 = spec.n2p if comb_w is None else 1
ws = model.workspace(batch, spec.nd, spec.n2p)
xs = torch.empty((batch, model.total), dtype=torch.float32, device='cuda')
history = torch.zeros(K, device='cuda')


def chunk(stream=None):
    model.net.fit_steps(solver.program, model.flat, xs, columns, 1234, 0, solver.grads, ws, adam.exp_avg, adam.exp_avg_sq, adam.mask,
                        adam.step_count, 1, adam.lr, adam.betas, adam.eps, history, K, dir_cols=spec.dir_cols, n2=n2,
                        ic_const=model.kernel_ic_const(), stream=stream)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (REPS * K)


plain = timed(chunk)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    chunk(side.cuda_stream)
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    chunk(torch.cuda.current_stream().cuda_stream)
replay = timed(graph.replay)
print(f'{name} batch {batch}: stream launches {plain * 1e6:.2f} us/iteration ({1 / plain:.0f} it/s)   hipGraph replay {replay * 1e6:.2f} us/iteration '
      f'({1 / replay:.0f} it/s)   ({(plain / replay - 1) * 100:+.1f} %)')
