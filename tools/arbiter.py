""" Accuracy of the kernels on TRAINED models, arbitrated in fp64: relative L2 error of loss and
gradient tensors for the product and for the fp32 oracle ("ref32"), both against the fp64 oracle, at the SAME trained
parameters. usage: python tools/arbiter.py [lib1.so lib2.so ...] [cfg ...]   (no library: the product build;
experiment builds carry the training kernels only, so the field is compared for the product build alone). """
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import pinn_configs as pc
import pydens_amd as pa
from pydens_amd import engine
from oracle import pinn_oracle as po
from helpers import export_grads, export_params, load_params, make_solver

args = sys.argv[1:]
libs = [a for a in args if a.endswith('.so')]
want = [a for a in args if not a.endswith('.so')] or ['cfg2', 'cfg4', 'cfg3', 'cfg1']
for name, iters, batch in (('cfg2', 600, 4096), ('cfg4', 400, 4096), ('cfg3', 150, 2048), ('cfg1', 1500, 100)):
    if name not in want:
        continue
    torch.manual_seed(13)
    cfg, trainer = make_solver(name, pa, gemm='fp32')
    sampler = pa.NumpySampler('uniform') & pa.NumpySampler('uniform', low=1, high=5) if name == 'cfg4' else None
    trainer.fit(niters=iters, batch_size=batch, sampler=sampler, lr=0.005)
    params = export_params(trainer)
    ocfg = pc.make_config(name, po.D, torch)
    pts = pc.sample_points(cfg, 4096, seed=17)
    ev = {}
    for dtype in (torch.float32, torch.float64):
        o = po.OracleSolver(ocfg['equation'], **ocfg['solver_kwargs'], dtype=dtype)
        o.import_params(params)
        e = o.evaluate(pts, chunk=2048)
        ev[dtype] = (e['loss'], o.export_grads(), e['u'])
    (l32, g32, u32), (l64, g64, u64) = ev[torch.float32], ev[torch.float64]
    ref = [np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(g32, g64) if b is not None]
    print(f'{name}: trained loss {float(trainer.losses[0]):.4g} -> {l64:.4g}; ref32 vs f64: loss {abs(l32 - l64) / l64:.2e}, '
          f'gradients median {np.median(ref):.2e} max {np.max(ref):.2e}', flush=True)
    flat = lambda ts: np.concatenate([np.asarray(t, dtype=np.float64).ravel() for t, b in zip(ts, g64) if b is not None])
    ref_all = np.linalg.norm(flat(g32) - flat(g64))
    for path, gemm in [(p, m) for p in (libs or [None]) for m in (('fp32', 'bf16x3') if name in ('cfg2', 'cfg4') else ('fp32',))]:
        solver = trainer
        if path is not None:
            _, solver = make_solver(name, pa, _lib=engine.bind(ctypes.CDLL(path)))
            load_params(solver, params)
        solver.set_gemm_mode(gemm)
        solver._fused_step(torch.from_numpy(pts).cuda(), 1)
        lay = solver.model.net.layout
        ours = [np.linalg.norm(g - b) / np.linalg.norm(b) for g, b in zip(export_grads(solver), g64) if b is not None]
        ratio = [a / b for a, b in zip(ours, ref)]
        loss = float(solver.grads[lay.off_loss])
        all_ratio = np.linalg.norm(flat(export_grads(solver)) - flat(g64)) / ref_all
        line = (f'   {os.path.basename(path) if path else "product":20s} {gemm:7s} all-tensors ours/ref32 {all_ratio:.2f}; loss {abs(loss - l64) / l64:.2e}, gradients median '
                f'{np.median(ours):.2e} max {np.max(ours):.2e}; ours/ref32 median {np.median(ratio):.2f} max {np.max(ratio):.2f}')
        if path is None and gemm == 'fp32':
            u = solver.predict(*[pts[:, c] for c in range(pts.shape[1])])
            line += f'; field {np.abs(u - u64).max() / np.abs(u64).max():.2e} (ref32 {np.abs(u32 - u64).max() / np.abs(u64).max():.2e})'
        print(line, flush=True)
