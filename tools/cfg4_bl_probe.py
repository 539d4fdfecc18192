""" Where does the error of dL/db_L on BASELINE config 4 come from? (VERDICT r5 "weak" 1: 1.15e-5 from the fp64 arbiter at
bench.parity_check's 4 096 points, seed 99 -- 2.26x the fp32 reference's own error.)

    g_bL = 2/N sum_p r_p G'(x_p),   r = G' net + G net_x - e pi cos(e pi x),   G(x) = sigmoid(x e^{-s}) - 1/2

a cancelling sum (r changes sign with the source term). The probe splits the kernel's error into its possible sources, all sums in fp64:
the kernel's per-point streams (pinn_jet_forward) against the fp64 oracle's, the source term in fp32 forms, the final summation.
usage: python tools/cfg4_bl_probe.py [seed ...] """
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np      # noqa: E402
import torch            # noqa: E402
import pinn_configs as pc       # noqa: E402
import pydens_amd as pa         # noqa: E402
from oracle import pinn_oracle as po    # noqa: E402
from helpers import export_grads, export_params     # noqa: E402

seeds = [int(a) for a in sys.argv[1:]] or [99, 100, 101]
torch.manual_seed(0)
cfg = pc.make_config('cfg4', pa.D, torch)
solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], device=torch.device('cuda', 0))
params = export_params(solver)
ocfg = pc.make_config('cfg4', po.D, torch)
n = 4096
for gemm in ('fp32', 'bf16x3'):
    solver.set_gemm_mode(gemm)
    for seed in seeds:
        pts = pc.sample_points(cfg, n, seed=seed)
        ev = {}
        for dtype in (torch.float32, torch.float64):
            o = po.OracleSolver(ocfg['equation'], **ocfg['solver_kwargs'], dtype=dtype)
            o.import_params(params)
            e = o.evaluate(pts, chunk=2048)
            ev[dtype] = (e, o.export_grads())
        (e32, g32), (e64, g64) = ev[torch.float32], ev[torch.float64]
        xs = torch.from_numpy(pts).cuda()
        solver.grads.zero_()
        solver._fused_step(xs, 1)
        torch.cuda.synchronize()
        ours = export_grads(solver)
        x, e = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
        s = float(params[-1])
        sg = 1.0 / (1.0 + np.exp(-x * np.exp(-s)))
        G, G1 = sg - 0.5, np.exp(-s) * sg * (1.0 - sg)
        src64 = e * np.pi * np.cos(e * np.pi * x)
        r64 = e64['r'][:, 0].astype(np.float64)
        r32 = e32['r'][:, 0].astype(np.float64)
        truth = 2.0 / n * np.sum(r64 * G1)
        # the kernel's streams (u, u_x) per point, fp32
        st = solver.model.net.jet_forward(solver.model.flat, xs, solver.spec.dir_cols, solver.spec.n2p,
                                          ic_const=solver.model.kernel_ic_const()).cpu().numpy().astype(np.float64)
        ux_k = st[1]
        ux64 = r64 + src64
        x32, e32_ = pts[:, 0], pts[:, 1]
        src32_ref = (e32_ * np.float32(np.pi) * np.cos(e32_ * np.float32(np.pi) * x32)).astype(np.float64)     # the reference's association, fp32
        cases = {
            'truth (fp64 oracle)': truth,
            'ref32 oracle gradient': float(g32[-2]),
            'kernel gradient': float(ours[-2]),
            'sum of ref32 r * G1_64': 2.0 / n * np.sum(r32 * G1),
            'kernel u_x - src64': 2.0 / n * np.sum((ux_k - src64) * G1),
            'kernel u_x - src32(ref order)': 2.0 / n * np.sum((ux_k - src32_ref) * G1),
            'u_x64 - src32(ref order)': 2.0 / n * np.sum((ux64 - src32_ref) * G1),
        }
        print(f'--- {gemm} seed {seed}: g_bL truth {truth:.9g}, sum |terms| {2.0 / n * np.sum(np.abs(r64 * G1)):.4g}')
        for k, v in cases.items():
            print(f'    {k:34s} {v:+.9e}  rel err {abs(v - truth) / abs(truth):.2e}')
        du = ux_k - ux64
        print(f'    kernel u_x error: mean {du.mean():+.2e} rms {np.sqrt((du ** 2).mean()):.2e} max {np.abs(du).max():.2e}; weighted bias sum(du G1)/sum(r G1) {np.sum(du * G1) / np.sum(r64 * G1):+.2e}')
        dr = r32 - r64
        print(f'    ref32  r  error: mean {dr.mean():+.2e} rms {np.sqrt((dr ** 2).mean()):.2e} max {np.abs(dr).max():.2e}; weighted bias {np.sum(dr * G1) / np.sum(r64 * G1):+.2e}')
        rel = [float(np.linalg.norm(np.asarray(a, dtype=np.float64) - np.asarray(c, dtype=np.float64)) / np.linalg.norm(np.asarray(c, dtype=np.float64)))
               for a, c in zip(ours, g64) if c is not None]
        rel32 = [float(np.linalg.norm(np.asarray(a, dtype=np.float64) - np.asarray(c, dtype=np.float64)) / np.linalg.norm(np.asarray(c, dtype=np.float64)))
                 for a, c in zip(g32, g64) if c is not None]
        print('    per tensor ours : ' + ' '.join(f'{v:.1e}' for v in rel))
        print('    per tensor ref32: ' + ' '.join(f'{v:.1e}' for v in rel32))
