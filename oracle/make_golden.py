""" TEST INFRASTRUCTURE ONLY -- generates `tests/golden/*.npz` from the UNMODIFIED reference.

Run in the build container (needs /root/reference): `python -m oracle.make_golden`.
For every workload in `pinn_configs.py` it builds the reference `Solver` (pydens/model_torch.py imported by
path with the batchflow stand-in), records its initial parameters, a fixed stream of point batches, and what
the reference computes on them: u_hat, residual, loss and parameter gradients on batch 0, then the losses and
final parameters of K Adam steps (`Solver.fit`, lr 0.005). These fixtures pin the oracle restatement
(`oracle/pinn_oracle.py`) and, on the GPU box where /root/reference is absent, the HIP engine.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import pinn_configs as pc                                   # noqa: E402
from oracle.reference_loader import load_reference          # noqa: E402

GOLDEN_N = dict(cfg1=100, cfg2=256, cfg3=128, cfg4=256, cfg5=64, ode_sigmoid=128, mixed=128, heat3d=128, kdv=128, resnet3=128,
                nested_acts=128, mixed3=128, biharm=128,     # round 5 breadth fixtures (tests/test_golden_extras.py)
                act_params=128,                              # round 6: activation instances with non-default parameters
                mixed31=128,                                 # round 6: u_xxxy, u_xyyy (weighted diagonals)
                mixed111=128)                                # round 6: u_xyz (three-column directions)
K_STEPS = 5
LR = 0.005


class FixedBatches:
    """ sampler plug-in (reference model_torch.py:433): returns pre-drawn batches in order. """
    def __init__(self, batches):
        self.batches, self.i = batches, 0

    def sample(self, size):
        out = self.batches[self.i]
        assert out.shape[0] == size
        self.i += 1
        return out.astype(np.float64)


def export(model):
    lins = [m for m in model.conv_block if isinstance(m, torch.nn.Linear)]
    params = [p for lin in lins for p in (lin.weight, lin.bias)] + [model.log_scale]
    return params


def main():
    ref = load_reference()
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)
    only = sys.argv[1:]                                      # `python -m oracle.make_golden mixed`: just these fixtures
    for idx, name in enumerate(GOLDEN_N):
        if only and name not in only:
            continue
        torch.manual_seed(100 + idx)
        cfg = pc.make_config(name, ref.D, torch)
        solver = ref.Solver(cfg['equation'], **cfg['solver_kwargs'])
        params = export(solver.model)
        n = GOLDEN_N[name]
        points = pc.sample_points(cfg, n, seed=idx, steps=K_STEPS)
        blob = {f'param_{i}': p.detach().numpy().copy() for i, p in enumerate(params)}
        blob['n_param_tensors'] = np.array(len(params))
        blob['points'] = points

        # evaluation on batch 0 with the initial parameters (reference model_torch.py:430-460 minus the step)
        xs = [torch.from_numpy(points[0][:, i:i + 1].copy()).requires_grad_() for i in range(points.shape[2])]
        u_hat = solver.ctx.run(solver.model, solver.reshape_and_concat(xs))
        r = solver.ctx.run(solver.equation, u_hat, *xs)
        loss = torch.nn.MSELoss()(r, torch.zeros_like(xs[0]))
        for p in params:
            p.grad = None
        loss.backward()
        blob['u_hat'] = u_hat.detach().numpy()[:, 0]
        blob['residual'] = r.detach().numpy()[:, 0]
        blob['loss0'] = loss.detach().numpy()
        for i, p in enumerate(params):
            blob[f'grad_{i}'] = np.zeros(0, dtype=np.float32) if p.grad is None else p.grad.numpy().copy()
        blob['predict'] = solver.predict(*[points[1][:, i] for i in range(points.shape[2])])[:, 0]

        # K Adam steps through the reference's own fit loop
        solver.fit(niters=K_STEPS, batch_size=n, sampler=FixedBatches(points), lr=LR)
        blob['losses'] = np.array([float(v) for v in solver.losses], dtype=np.float32)
        for i, p in enumerate(params):
            blob[f'final_{i}'] = p.detach().numpy().copy()
        blob['lr'] = np.array(LR)
        path = os.path.join(out_dir, f'{name}.npz')
        np.savez(path, **blob)
        print(name, 'loss0', float(blob['loss0']), 'losses', blob['losses'], os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
