""" Run-to-run repeatability of the fused step of one build of the library: the same parameters and points, `--reps` steps, the
loss, |g|_1 and a hash of the whole gradient buffer of every step, at a cap of the workgroups per CU (pinn_debug_max_wgs_per_cu).
Usage: python tools/repeat_check.py cfg2,cfg4 lib1.so,lib2.so [--caps 0,1] [--gemms fp32,bf16x3] [--n 65536] [--reps 6]
(the probe behind DESIGN.md's account of the two-workgroups-per-CU finding; tools/var2.sh builds the libraries) """
import argparse
import ctypes
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
import pinn_configs as pc   # noqa: E402
import pydens_amd as pa     # noqa: E402
from pydens_amd import engine   # noqa: E402


def check(workload, lib_path, cap, gemm, n, reps):
    lib = engine.bind(ctypes.CDLL(lib_path))
    torch.manual_seed(0)
    cfg = pc.make_config(workload, pa.D, torch, V=pa.V)
    solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], _lib=lib)
    lib.pinn_debug_max_wgs_per_cu(solver.model.net.handle, cap)
    solver.set_gemm_mode(gemm)
    n = n or min(cfg['n_points'], 131072)
    xs = torch.from_numpy(pc.sample_points(cfg, n, seed=1)).cuda()
    lay = solver.model.net.layout
    canary = None
    if hasattr(lib, 'pinn_debug_phase_buffer') and os.environ.get('PINN_CANARY'):       # -DPINN_SCRATCH_CANARY builds
        canary = torch.zeros(64, dtype=torch.int64, device='cuda')
        lib.pinn_debug_phase_buffer(ctypes.c_void_p(canary.data_ptr()))
    seen = []
    for r in range(reps):
        solver.grads.zero_()
        solver._fused_step(xs, 1)
        torch.cuda.synchronize()
        g = solver.grads[:lay.p_total].cpu().numpy()
        seen.append((float(g[lay.off_loss]), float(np.abs(g[:lay.p_core].astype(np.float64)).sum()), hashlib.sha1(g.tobytes()).hexdigest()[:12]))
    info = (ctypes.c_int32 * 4)()
    lib.pinn_last_launch_info(info)
    distinct = len({s[2] for s in seen})
    print(f'{os.path.basename(lib_path):22s} {workload} {gemm:6s} n={n} cap={cap} {lib.pinn_last_kernel_name().decode()} '
          f'grid {info[0]} = {info[1]}/CU x {info[2]} threads, lds {info[3]}: {distinct} distinct gradient buffer(s) in {reps} runs', flush=True)
    for loss, g1, h in seen:
        print(f'    loss {loss:.7f}  |g|1 {g1:.7f}  sha1 {h}')
    if canary is not None:
        c = canary.cpu().numpy()
        print(f'    scratch canary: {c[0]} tiles read back a tag that is not their own'
              + (f' (first: wanted block {c[1] >> 12} thread {c[1] & 4095}, got block {c[2] >> 12} thread {c[2] & 4095} = {int(c[2]):#x})' if c[0] else ''))
        lib.pinn_debug_phase_buffer(None)
    lib.pinn_debug_max_wgs_per_cu(solver.model.net.handle, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('workloads', help='comma-separated: cfg2,cfg4,...')
    ap.add_argument('libs', help='comma-separated library paths')
    ap.add_argument('--caps', default='0')
    ap.add_argument('--gemms', default='fp32')
    ap.add_argument('--n', type=int, default=None)
    ap.add_argument('--reps', type=int, default=6)
    args = ap.parse_args()
    for lib_path in args.libs.split(','):
        for gemm in args.gemms.split(','):
            for workload in args.workloads.split(','):
                for cap in args.caps.split(','):
                    check(workload, lib_path, int(cap), gemm, args.n, args.reps)


if __name__ == '__main__':
    main()
