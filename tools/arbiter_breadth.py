""" Accuracy of the breadth kernels on TRAINED models, arbitrated in fp64, like tools/arbiter.py does for the BASELINE
shapes: per parameter tensor the relative L2 error of the gradient for the product (or experiment builds given as .so paths) and for the
fp32 oracle, both against the fp64 oracle, after `iters` Adam steps.  usage: python tools/arbiter_breadth.py [lib.so ...] [sin64 program skip128] """
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import pinn_configs as pc
import pydens_amd as pa
from pydens_amd import engine
from oracle import pinn_oracle as po
from helpers import export_grads, export_params, load_params

args = sys.argv[1:]
libs = [a for a in args if a.endswith('.so')] or [None]
want = [a for a in args if not a.endswith('.so')] or ['sin64', 'program', 'skip128']
for name in want:
    for iters in (0, 150, 600):
        torch.manual_seed(13)
        cfg = pc.make_config(name, pa.D, torch, V=pa.V)
        trainer = pa.Solver(cfg['equation'], **cfg['solver_kwargs'])
        if iters:
            trainer.fit(niters=iters, batch_size=4096, lr=0.005)
        params = export_params(trainer)
        ocfg = pc.make_config(name, po.D, torch, V=po.V)
        pts = pc.sample_points(cfg, 4096, seed=17)
        ev = {}
        for dtype in (torch.float32, torch.float64):
            o = po.OracleSolver(ocfg['equation'], **ocfg['solver_kwargs'], dtype=dtype)
            o.import_params(params)
            for vname in getattr(trainer.model, 'variables', {}):
                if hasattr(o.model, vname):
                    getattr(o.model, vname).data.copy_(getattr(trainer.model, vname).detach().cpu().to(dtype))
            e = o.evaluate(pts, chunk=2048)
            ev[dtype] = (e['loss'], o.export_grads())
        (l32, g32), (l64, g64) = ev[torch.float32], ev[torch.float64]
        flat = lambda ts: np.concatenate([np.asarray(t, dtype=np.float64).ravel() for t, b in zip(ts, g64) if b is not None])
        ref_all = np.linalg.norm(flat(g32) - flat(g64)) / np.linalg.norm(flat(g64))
        print(f'{name} after {iters} steps: loss {l64:.4g}; ref32 vs f64: loss {abs(l32 - l64) / l64:.2e}, all gradients {ref_all:.2e}', flush=True)
        for path in libs:
            solver = trainer
            if path is not None:
                solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], _lib=engine.bind(ctypes.CDLL(path)))
                load_params(solver, params)
                for vname in getattr(trainer.model, 'variables', {}):
                    getattr(solver.model, vname).data.copy_(getattr(trainer.model, vname).detach())
            solver._fused_step(torch.from_numpy(pts).cuda(), 1)
            lay = solver.model.net.layout
            ours_all = np.linalg.norm(flat(export_grads(solver)) - flat(g64)) / np.linalg.norm(flat(g64))
            loss = float(solver.grads[lay.off_loss])
            print(f'   {os.path.basename(path) if path else "product":24s} loss {abs(loss - l64) / l64:.2e}, all gradients {ours_all:.2e} = {ours_all / ref_all:.2f} x ref32 '
                  f'({solver.model.net.lib.pinn_last_kernel_name().decode()})', flush=True)
