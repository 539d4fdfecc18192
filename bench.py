""" bench.py -- collocation-points/sec of the full residual + grad + Adam step (BASELINE.json metric).

Default workload: BASELINE config 2 (2D Poisson, 4x64 Tanh MLP, 65 536 points PER GPU, weak scaling) -- the configuration
the metric is quoted on. `--workload cfg3|cfg4|cfg5` times the other BASELINE configs the same way (cfg3: 262 144 points,
cfg4 / cfg5: 131 072 points per GPU = their 1 048 576-point step on 8 GPUs); `--global-points N` fixes the GLOBAL batch
instead (strong scaling: N / n_gpus points per GPU -- north_star's "1M points per step on 8 GPUs").

A "step" = one fused pinn_residual_step (forward jets, ansatz, residual, MSE, reverse sweep; for widths >= 128 the
weight gradients come from the streamed pinn_wgrad_kernel) + gradient all-reduce over RCCL (N > 1) + Adam, on point
batches already resident in HBM; it is the very iteration `Solver.fit` runs (Solver._dp_step for N > 1).

    python bench.py --gpus 1 --steps 100 --warmup 20 [--workload cfg2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0). Timed region: K steps between two HIP events recorded on the launch stream, bracketed
by a barrier + device synchronise on both sides; the MAX over ranks is reported. `roofline`: algorithmic FLOPs of the
step's matrix kernels per launch = F * points, F = 6*S*sum_{l>=2}(in*out) + 4*d*H1, over their mean
duration measured with HIP events on the launch stream (pinn_profile_tile), against the fp32 MFMA peak of MI355X
(157.3 TFLOP/s). `cpu_baseline`: the oracle restatement of the reference step (oracle/pinn_oracle.py, torch CPU ops)
timed in the same run on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import contextlib
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np      # noqa: E402
import torch            # noqa: E402

import pinn_configs as pc       # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # ... "Peak BF16/FP16 MFMA", dense
POOL = 8                        # distinct pre-generated point batches cycled through
# per-GPU batch, CPU-baseline batch (min(N_cfg, 65 536), 16 384 for cfg3 / cfg5), description
WORKLOADS = {
    'cfg2': (65536, 65536, '2D Poisson u_xx+u_yy=5sin(pi(x+y)), BC=1, 4x64 Tanh MLP'),
    'cfg3': (262144, 16384, '3D heat u_xx+u_yy=u_t (x,y,t), IC 10xy(1-x)(1-y), BC=0, 5x128 Tanh MLP'),
    'cfg4': (131072, 65536, 'parametric ODE u_x=e*pi*cos(e*pi*x) (x,e), IC=1, 4x64 Tanh MLP (1 048 576 points / 8 GPUs)'),
    'cfg5': (131072, 16384, '2D wave u_tt=u_xx (x,t), IC x(1-x), BC=0, 6x256 Tanh MLP (1 048 576 points / 8 GPUs)'),
    # breadth workloads (not BASELINE configs): the paths a user gets off the five BASELINE shapes
    'skip128': (65536, 16384, "breadth: 2D Poisson, layout 'faR fa fa+ fa f' (skip connection), 4x128 Tanh -- skip kernels (VAR 8 | 1024) + streamed weight gradients"),
    'skip256': (65536, 4096, "breadth: 2D Poisson, layout 'faR fa fa+ R fa fa+ f' (two residual blocks), 5x256 Tanh -- skip kernels + streamed weight gradients"),
    'sin128': (65536, 16384, "breadth: 2D Poisson, 4x128 MLP with activation 'Sin' -- full breadth kernel (VAR 8 | 128) + streamed weight gradients"),
    'gelu256': (65536, 4096, "breadth: Burgers u_t+u u_x=0.05 u_xx (IC sin(pi x), BC 0), layout 'fa fRa fa f+a f', 4x256 GELU -- full breadth kernel + residual program + streamed weight gradients"),
    'sin64': (65536, 16384, "breadth: 2D Poisson, 4x64 MLP with activation 'Sin' -- breadth kernels (VAR 8)"),
    'program': (65536, 16384, 'breadth: u_xx+u_yy+k u^2=5sin(pi(x+y)) with a trainable V(k), 4x64 Tanh -- residual program + its reverse sweep'),
    'generic': (65536, 16384, 'breadth: BASELINE cfg2 on the GENERIC step path (pinn_jet_forward -> torch autograd over the equation -> pinn_jet_backward)'),
    # round 6
    'burgers64': (65536, 16384, 'breadth: viscous Burgers u_t+u u_x=0.05 u_xx (x,t), IC sin(pi x), BC 0, 4x64 Tanh -- evolution shape with a residual program on the two-team kernel (VAR 256 | 32 | 2048)'),
    'heat64': (65536, 16384, 'breadth: 1-D heat u_t=0.3 u_xx+2e^{-t}sin(pi x) (x,t), IC sin(pi x), BC 0, 4x64 Tanh -- evolution shape, affine residual, two-team kernel (VAR 256 | 32)'),
    'poisson512': (65536, 4096, 'breadth: BASELINE cfg2 problem on a 4x512 Tanh MLP -- hidden width 512: generic step path, one kernel call of S = 3 streams per second-order direction, weight gradients in four 256 x 256 block passes'),
}
BASELINE_WORKLOADS = ('cfg2', 'cfg3', 'cfg4', 'cfg5')


def flops_per_point(layer_dims, n_streams):
    d, h1 = layer_dims[0], layer_dims[1]
    inner = sum(a * b for a, b in zip(layer_dims[1:-1], layer_dims[2:]))
    return 6 * n_streams * inner + 4 * d * h1


def cpu_baseline(workload, n_points, budget_s=15.0):
    """ oracle (port of the reference step: oracle/pinn_oracle.py restates pydens Solver.fit op for op and is pinned
    bit-level to fixtures generated by the unmodified reference, tests/test_oracle_vs_golden.py; /root/reference itself
    does not exist on the GPU box) on the host cores; bounded sample of the same workload: batches of `n_points` points
    (BASELINE.md section 3.3), as many Solver.fit iterations as fit in ~budget_s seconds. The intra-op thread count is
    tuned first (ATen's CPU ops on [N,64] tensors slow down when oversubscribed) and reported as `cores`. """
    from oracle import pinn_oracle as po
    cfg = pc.make_config(workload, po.D, torch, V=po.V)
    pts = pc.sample_points(cfg, n_points, seed=0, steps=1)
    ncpu = os.cpu_count() or 1
    best = None
    for threads in sorted({min(t, ncpu) for t in (8, 16, 32)}):
        torch.set_num_threads(threads)
        solver = po.OracleSolver(cfg['equation'], **cfg['solver_kwargs'])
        solver.fit(niters=1, batch_size=n_points, points=pts)            # warm-up
        t0 = time.perf_counter()
        solver.fit(niters=1, batch_size=n_points, points=pts)
        per_step = time.perf_counter() - t0
        if best is None or per_step < best[1]:
            best = (threads, per_step)
    threads, per_step = best
    torch.set_num_threads(threads)
    steps = int(max(3, min(400, budget_s / max(per_step, 1e-3))))
    print(f'[bench] cpu baseline: {threads} threads ({ncpu} logical CPUs), {per_step * 1e3:.1f} ms/step, '
          f'timing {steps} steps', file=sys.stderr)
    solver = po.OracleSolver(cfg['equation'], **cfg['solver_kwargs'])
    solver.fit(niters=1, batch_size=n_points, points=pts)
    t0 = time.perf_counter()
    solver.fit(niters=steps, batch_size=n_points, points=np.repeat(pts, steps, axis=0))
    dt = time.perf_counter() - t0
    return dict(value=n_points * steps / dt, unit='points/s', cores=threads, kind='port', n_cpu=n_points,
                sample=f'{steps} Solver.fit iterations of {workload} at batch {n_points} (N_cpu of BASELINE.md 3.3) by the '
                       f'PORT oracle/pinn_oracle.py (reference step restated op for op, pinned to reference-generated '
                       f'fixtures; torch {torch.__version__} CPU ops, fp32; best of 8/16/32 intra-op threads on {ncpu} '
                       f'logical CPUs), {dt:.1f} s')


def nn_mse():
    import torch.nn as nn
    return nn.MSELoss()


_FROZEN = []


def freeze_once():
    """ once per process (main calls it first): what is alive now (torch, the modules) goes to the collector's permanent generation -- later
    collections stay short, and no timed region is preceded by 40 ms of idle GPU (which would cost the clocks the settle load has brought up) """
    import gc
    if not _FROZEN:
        gc.collect()
        gc.freeze()
        _FROZEN.append(True)


@contextlib.contextmanager
def quiet_collector():
    """ timed regions run with Python's cyclic garbage collector paused, as `timeit` does: a full collection in a process that has torch
    loaded stops the launching thread for ~40 ms (measured on `Solver.fit` of BASELINE config 4, round 6: tools/fit_one.py), which the GPU
    then spends idle -- a property of the host interpreter at the moment of the measurement, not of the step """
    import gc
    freeze_once()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def parity_check(workload, solver, generic, mse, n_points=4096, seed=99, cache=None):
    """ OUTSIDE the timed region: the step path of the run, on `n_points` fresh points at the solver's current parameters,
    against the oracle (the checker: SURVEY 8c) -- loss within 1e-5, every parameter gradient within 1e-5 relative L2 (fp64 arbiter
    where the fp32 reference is the noisy side). A bench line of a breadth workload says with it that the number belongs to a kernel
    that computes the right thing. """
    import pinn_configs as pc_
    import pydens_amd as pa_
    from oracle import pinn_oracle as po
    ocfg = pc_.make_config(workload, po.D, torch, V=po.V)
    lins = [m for m in solver.model.conv_block]
    params = []
    for lin in lins:
        params += [lin.weight.detach().cpu().numpy().copy(), lin.bias.detach().cpu().numpy().copy()]
    params.append(solver.model.log_scale.detach().cpu().numpy().copy())
    pts = pc_.sample_points(pc_.make_config(workload, pa_.D, torch, V=pa_.V), n_points, seed=seed)
    worst, loss_err = 0.0, 0.0
    # (`cache`: the oracle's answers are a function of workload, points and parameters only -- tests that check several GEMM modes at
    #  the same parameters evaluate it once)
    import hashlib
    key = (workload, n_points, seed, hashlib.sha1(b''.join(np.ascontiguousarray(p).tobytes() for p in params)).hexdigest())
    refs = cache.get(key) if cache is not None else None
    if refs is None:
        refs = {}
        for dtype in (torch.float32, torch.float64):
            oracle = po.OracleSolver(ocfg['equation'], dtype=dtype, **ocfg['solver_kwargs'])
            oracle.import_params(params)
            for name in getattr(solver.model, 'variables', {}):          # trainable V(...) scalars of the equation
                if hasattr(oracle.model, name):
                    getattr(oracle.model, name).data.copy_(getattr(solver.model, name).detach().cpu().to(dtype))
            ev = oracle.evaluate(pts, chunk=2048)
            refs[dtype] = (ev['loss'], oracle.export_grads())
        if cache is not None:
            cache[key] = refs
    xs = torch.from_numpy(pts).to(solver.model.flat.device)
    solver.grads.zero_()
    if generic:
        solver._generic_step(xs, ('equation',), [], mse, 1)
    else:
        solver._fused_step(xs, 1)
    torch.cuda.synchronize()
    lay = solver.model.net.layout
    loss = float(solver.grads[lay.off_loss])
    (l32, g32), (l64, g64) = refs[torch.float32], refs[torch.float64]
    ok = abs(loss - l64) <= max(2 * abs(l32 - l64), 1e-5 * abs(l64))
    loss_err = abs(loss - l64) / abs(l64)
    got = []
    for w, b in solver.model.net.param_views(solver.grads):
        got += [w.detach().cpu().numpy(), b.detach().cpu().numpy()]
    got.append(solver.grads[lay.off_log_scale].detach().cpu().numpy())
    per_tensor = []
    for g, a32, a64 in zip(got, g32, g64):
        if a64 is None:
            continue
        a64 = np.asarray(a64, dtype=np.float64)
        err = float(np.linalg.norm(np.asarray(g, dtype=np.float64) - a64))
        ref = float(np.linalg.norm(np.asarray(a32, dtype=np.float64) - a64))
        scale = float(np.linalg.norm(a64))
        worst = max(worst, err / max(scale, 1e-30))
        per_tensor.append([float('%.2e' % (err / max(scale, 1e-30))), float('%.2e' % (ref / max(scale, 1e-30)))])
        ok = ok and err <= max(2 * ref, 1e-5 * scale + 1e-9 * np.sqrt(a64.size))
    return {'ok': bool(ok), 'points': n_points, 'seed': seed, 'loss_rel_err': loss_err, 'worst_gradient_rel_err': worst,
            'gradient_rel_err_per_tensor_ours_ref32': per_tensor,
            'against': 'oracle/pinn_oracle.py (fp32; fp64 arbiter, SURVEY 8c item 5), outside the timed region'}


def executed_streams(solver, generic=False):
    """ streams the matrix pipe executes: the second derivatives of a Laplace / heat / wave operator travel as ONE combined stream """
    plan, spec = solver.residual_plan, solver.spec
    return (2 + spec.nd) if (plan is not None and plan.comb_w is not None and not generic) else spec.n_streams


def baseline_configs(device, no_parity=False, steps=20, warmup=5):
    """ compact figures of the BASELINE configs the headline is NOT quoted on, in the same process and on the same GPU: cfg3, cfg4, cfg5 at
    their per-GPU batch (`steps` timed steps of the single-rank step path between two HIP events on the launch stream after `warmup` -- the
    GPU is in its steady clock state, they run behind the headline's timed region --, matrix-kernel durations by HIP events around the
    launches, parity_check against the oracle) and cfg1 as `Solver.fit` iterations / s at its 100 points (the reference's own regime). """
    import pydens_amd as pa
    from pydens_amd.solver import FlatAdam
    out = {}
    for wl in ('cfg3', 'cfg4', 'cfg5'):
        n = WORKLOADS[wl][0]
        torch.manual_seed(0)
        cfg = pc.make_config(wl, pa.D, torch, V=pa.V)
        solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], device=device)
        model, spec = solver.model, solver.spec
        parity = None if no_parity else parity_check(wl, solver, False, None)
        solver.optimizer = FlatAdam(model, lr=0.005)
        solver.optimizer.refresh()
        stream = pa.engine.stream_of(model.flat)
        lo = torch.tensor(cfg['low'], dtype=torch.float32, device=device)
        hi = torch.tensor(cfg['high'], dtype=torch.float32, device=device)
        gen = torch.Generator(device=device)
        gen.manual_seed(4321)
        pool = [(lo + (hi - lo) * torch.rand((n, model.total), dtype=torch.float32, device=device, generator=gen)).contiguous() for _ in range(2)]
        for i in range(warmup):
            solver._fused_step(pool[i % 2], 1, adam=solver.optimizer, stream=stream)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # (round 6: a step of cfg4 is 0.2 ms -- 5 + 20 of them behind the oracle's seconds of CPU work ran in the clock state of an idle GPU,
        #  8 % above the workload's own bench line; every config now warms up for >= 30 ms and is timed over >= 60 ms, at least `steps` steps)
        torch.cuda.synchronize()
        ev0.record()
        solver._fused_step(pool[0], 1, adam=solver.optimizer, stream=stream)
        ev1.record()
        torch.cuda.synchronize()
        est = max(ev0.elapsed_time(ev1), 1e-3)
        steps_wl = max(steps, int(np.ceil(60.0 / est)))
        for i in range(max(0, int(np.ceil(30.0 / est)) - warmup)):
            solver._fused_step(pool[i % 2], 1, adam=solver.optimizer, stream=stream)
        torch.cuda.synchronize()
        ev0.record()
        for i in range(steps_wl):
            solver._fused_step(pool[i % 2], 1, adam=solver.optimizer, stream=stream)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / steps_wl
        lib = model.net.lib
        lib.pinn_profile_tile(1)
        t_tile, t_wg = [], []
        for i in range(min(steps_wl, 40)):
            solver._fused_step(pool[i % 2], 1, stream=stream)
            t_tile.append(float(lib.pinn_last_tile_ms()))
            t_wg.append(float(lib.pinn_last_wgrad_ms()))
        lib.pinn_profile_tile(0)
        kernels_ms = float(np.mean(t_tile)) + (float(np.mean(t_wg)) if min(t_wg) >= 0 else 0.0)
        f_pt, f_exec = flops_per_point(model.layer_dims, spec.n_streams), flops_per_point(model.layer_dims, executed_streams(solver))
        entry = {'points_per_gpu': n, 'steps': steps_wl, 'ms_per_step': ms, 'value': n / (ms * 1e-3), 'kernel_ms': kernels_ms,
                 'kernel': lib.pinn_last_kernel_name().decode(),
                 'frac': f_pt * n / (kernels_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                 'frac_executed': f_exec * n / (kernels_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                 'parity_ok': None if parity is None else parity['ok'],
                 'worst_gradient_rel_err': None if parity is None else parity['worst_gradient_rel_err']}
        out[wl] = entry
        del solver, pool
        torch.cuda.empty_cache()
    # cfg1: the reference's own CPU-runnable case -- `Solver.fit` end to end (device sampler, chunks of iterations as one launch)
    torch.manual_seed(0)
    cfg = pc.make_config('cfg1', pa.D, torch, V=pa.V)
    solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], device=device)
    parity = None if no_parity else parity_check('cfg1', solver, False, None, n_points=100)
    iters = 4096
    with quiet_collector():
        solver.fit(niters=512, batch_size=100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.fit(niters=iters, batch_size=100)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out['cfg1'] = {'points_per_gpu': 100, 'fit_iterations': iters, 'fit_it_per_s': iters / dt, 'us_per_iteration': dt / iters * 1e6,
                   'value': 100 * iters / dt, 'kernel': solver.model.net.lib.pinn_last_kernel_name().decode(),
                   'parity_ok': None if parity is None else parity['ok'],
                   'worst_gradient_rel_err': None if parity is None else parity['worst_gradient_rel_err'],
                   'what': 'Solver.fit(niters, batch_size=100) end to end by the host clock (device sampler, Adam, loss history): iterations / s'}
    return out


def hbm_traffic(workload, kernel_name, n_points):
    """ HBM-side bytes per launch of the dominant kernel from the COMMITTED rocprofv3 PMC passes of this same command
    (FETCH_SIZE / WRITE_SIZE, separate --pmc runs, gfx950 correction applied: the JSON says how) -- counters cannot be
    read from inside the run, so this is not a measurement of THIS run; None unless kernel and batch match. """
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r*_{workload}*_pmc.json')), reverse=True):     # newest round first
        try:
            with open(path) as f:
                pmc = json.load(f)
        except (OSError, ValueError):
            continue
        if pmc.get('kernel') == kernel_name and pmc.get('points_per_launch') == n_points:
            # ... and only if the profile was taken from the kernel sources of THIS tree (a kernel change that keeps the template
            # name must not inherit old bytes)
            from pydens_amd.csrc import build as hip_build
            if pmc.get('kernel_sources_sha1') != hip_build.kernel_sources_sha1():
                return None, f'{os.path.relpath(path, ROOT)} is from other kernel sources (stale): re-run tools/profile_bench.sh'
            return pmc['hbm_bytes_per_launch'], os.path.relpath(path, ROOT)
    return None, None


def kernel_variant(name):
    """ 'pinn_tile_kernel<64,2,1,1,3,0,true,784>' -> '512>' if the VAR bit of the split-bf16 kernels is set (so that a bf16x3 request
    that fell back to an fp32 kernel is not mislabelled) """
    try:
        var = int(name.rstrip('>').split(',')[-1])
    except ValueError:
        return ''
    return '512>' if var & 512 else ''


def wgrad_split(name):
    """ 'pinn_wgrad_kernel<HP,ND,N2,COMB,MT,SPLIT,SKIPS,HEAVY>' -> is SPLIT set? """
    fields = name.rstrip('>').split(',')
    return len(fields) > 5 and fields[5] == 'true'


def relaunch(n_gpus):
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print('[bench] no WORLD_SIZE in the environment: launching ' + ' '.join(cmd), file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='cfg2')
    ap.add_argument('--batch', type=int, default=None, help='points per GPU per step (default: the workload\'s)')
    ap.add_argument('--global-points', type=int, default=None,
                    help='strong scaling: GLOBAL points per step, split evenly over the GPUs')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--settle', type=float, default=0.3,
                    help='seconds of unrelated GEMM load before the warmup steps (GPU clock ramp of a fresh process); 0 = off')
    ap.add_argument('--lib', default=None, help='experiment build of the library to time instead of the product (tools/variant.sh)')
    ap.add_argument('--gemm', choices=('fp32', 'bf16x3'), default='fp32',
                    help='hidden-layer GEMM arithmetic: fp32 = exact-fp32 MFMA (the headline), bf16x3 = the gated split-bf16 variant '
                         '(fp32 operands as 3 x bf16, six products, fp32 accumulate; cfg2 / cfg4 kernels)')
    ap.add_argument('--tanh', choices=('fast', 'accurate'), default='fast',
                    help='Solver.set_tanh_mode: accurate = polynomial tanh below |z| = 0.45 (the cfg2 kernel; +2.5 % time, trained-state gradient '
                         'error at the level of the fp32 reference itself)')
    ap.add_argument('--device', choices=('cuda', 'cpu'), default='cuda',
                    help='cpu: plumbing tests only (needs --lib with the emulator build)')
    ap.add_argument('--no-side', action='store_true',
                    help='skip the side figure of the other GEMM mode (`gemm_bf16x3` key): tools/profile_bench.sh, so that a profile holds one set of kernels')
    ap.add_argument('--no-strong', action='store_true',
                    help='skip the strong-scaling side measurement (north_star: 1 048 576 points per step over all GPUs)')
    ap.add_argument('--no-parity', action='store_true',
                    help='skip the 4 096-point oracle comparisons before / after the timed steps (tools/profile_bench.sh: a rocprofv3 '
                         'kernel row then holds launches of ONE grid size -- the parity launches are 16 x smaller)')
    ap.add_argument('--no-cold', action='store_true',
                    help='skip the `cold` pass (the same W + K steps taken first, before the --settle load): tools/profile_bench.sh, so '
                         'that the average duration of a profile is the settled one')
    ap.add_argument('--no-configs', action='store_true',
                    help='skip the compact `baseline_configs` object of the default run (cfg1 fit rate, cfg3 / cfg4 / cfg5 step time, roofline '
                         'fractions and parity) and the `sustained` figure: tools/profile_bench.sh')
    ap.add_argument('--unfused', action='store_true',
                    help='diagnostic: run the N > 1 step path (step, all-reduce, Adam as separate launches) at N = 1')
    args = ap.parse_args()
    freeze_once()

    import pydens_amd as pa
    if args.lib:
        import ctypes
        from pydens_amd import engine
        engine._LIB = engine.bind(ctypes.CDLL(os.path.abspath(args.lib)))
        print(f'[bench] EXPERIMENT BUILD {args.lib}', file=sys.stderr)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU under torch.distributed.run on
        # 127.0.0.1, a free port); the ranks' stdout passes through, so rank 0's JSON line stays the last line of this process
        relaunch(args.gpus)
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE)')
    on_cpu = args.device == 'cpu'
    if on_cpu:
        # plumbing tests on a machine without a GPU (tests/test_bench_plumbing.py): the emulator build of the library behind the
        # same C-ABI, gloo instead of RCCL; nothing of it is a measurement
        if not args.lib:
            raise SystemExit('--device cpu needs --lib <emulator build of the library> (test plumbing only)')
        device = torch.device('cpu')
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
    dist = torch.distributed
    if world > 1 or args.unfused:
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
        if on_cpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)

    wl_batch, cpu_batch, wl_text = WORKLOADS[args.workload]
    big = args.workload in ('cfg3', 'cfg5')
    steps = args.steps if args.steps is not None else (20 if big else 100)
    warmup = args.warmup if args.warmup is not None else (5 if big else 20)
    if args.global_points:
        if args.global_points % world:
            raise SystemExit(f'--global-points {args.global_points} is not a multiple of {world} GPUs')
        n, scaling = args.global_points // world, 'strong'
    else:
        n, scaling = (args.batch or wl_batch), 'weak'

    torch.manual_seed(0)
    cfg = pc.make_config(args.workload, pa.D, torch, V=pa.V)
    extra = dict(_lib=pa.engine._LIB) if on_cpu else {}
    solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], device=device, **extra)
    generic = args.workload in ('generic', 'poisson512')     # (width 512: S <= 3 streams per call -- the Laplacian runs in direction groups on the generic path)
    assert generic or solver.program is not None, solver.program_error
    if generic:
        solver.program = None                   # what a user gets whose equation the tracer cannot lower
    solver.set_gemm_mode(args.gemm)
    solver.set_tanh_mode(args.tanh)
    model, spec = solver.model, solver.spec
    lay = model.net.layout
    lo = torch.tensor(cfg['low'], dtype=torch.float32, device=device)
    hi = torch.tensor(cfg['high'], dtype=torch.float32, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)

    def make_pool(points):
        return [(lo + (hi - lo) * torch.rand((points, model.total), dtype=torch.float32, device=device, generator=gen)).contiguous()
                for _ in range(POOL)]

    pool = make_pool(n)
    # BEFORE anything is timed: the step path of this run against the oracle on 4 096 points at the initial parameters (rank 0)
    parity_at_start = None
    if rank == 0 and not on_cpu and not args.lib and not args.no_parity:
        parity_at_start = parity_check(args.workload, solver, generic, nn_mse())

    from pydens_amd.solver import FlatAdam
    solver.optimizer = FlatAdam(model, lr=0.005)
    solver.optimizer.refresh()
    dp = world > 1 or args.unfused
    if dp:
        solver.begin_data_parallel()            # broadcast from rank 0, RCCL communicator on the compute stream
    stream = pa.engine.stream_of(model.flat)

    import torch.nn as nn
    mse = nn.MSELoss()

    def step(i, pts):
        if generic:
            solver._generic_step_auto(pts[i % POOL], ('equation',), [], mse, world)     # (what Solver.fit calls: launch-graph replay when it can)
            if dp:
                solver._all_reduce(stream)
            solver.optimizer.step(solver.grads, stream=stream)
        elif not dp:
            solver._fused_step(pts[i % POOL], 1, adam=solver.optimizer, stream=stream)   # Adam inside the reduction launch
        else:
            solver._dp_step(pts[i % POOL], world, stream=stream)

    def sync():
        if not on_cpu:
            torch.cuda.synchronize()

    per_rank = []           # seconds of the last timed() call, rank by rank

    def timed(pts, n_warm, n_steps):
        """ n_warm untimed steps, then EXACTLY n_steps between two HIP events on the launch stream, barrier + device
        synchronise on both sides; -> (seconds by the events, seconds by the host clock), MAX over the ranks """
        with quiet_collector():
            for i in range(n_warm):
                step(i, pts)
            if not on_cpu:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if world > 1:
                dist.barrier()
            sync()
            t0 = time.perf_counter()
            if not on_cpu:
                ev0.record()
            for i in range(n_steps):
                step(i, pts)
            if not on_cpu:
                ev1.record()
            if world > 1:
                dist.barrier()
            sync()
            host = time.perf_counter() - t0
        t = torch.tensor([host if on_cpu else ev0.elapsed_time(ev1) * 1e-3, host], dtype=torch.float64, device=device)
        if world > 1:
            # every rank's own figure too (rank 0 prints min / max: the first multi-GPU run should say WHERE a shortfall sits)
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            per_rank[:] = [float(e[0].item()) for e in every]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        else:
            per_rank[:] = [float(t[0].item())]
        return float(t[0].item()), float(t[1].item())

    # A fresh process finds the GPU in its idle power state: the first ~50 ms of load run 8-12 % slower than the steady state a
    # training run lives in (BASELINE config 2, same box: 5 + 20 steps 0.229 ms/step, 50 + 300 steps 0.2005, 100 + 1000 steps
    # 0.2000). Both are measured and printed: `cold` = W + K steps straight away (the power-state ramp inside the timed region),
    # then --settle seconds of dense fp32 GEMMs that have nothing to do with the workload (no kernel, buffer or parameter of the
    # step is touched; a memory-bound load does not raise the clocks), then W + K steps again = the steady state = `value`.
    # DESIGN.md section 6a.
    cold = None
    if args.settle > 0 and not on_cpu:
        if not args.no_cold:
            cold_dt, cold_host = timed(pool, warmup, steps)
            cold = {'ms_per_step': cold_dt / steps * 1e3, 'value': n * world * steps / cold_dt,
                    'what': f'the same {warmup} + {steps} steps taken first, in the fresh process, before the --settle load'}
        a = torch.randn(2048, 2048, device=device)
        b = torch.randn(2048, 2048, device=device)
        t_end = time.perf_counter() + args.settle
        while time.perf_counter() < t_end:
            for _ in range(20):
                a @ b
            torch.cuda.synchronize()
        del a, b
    dt, host_dt = timed(pool, warmup, steps)
    loss = float(solver.grads[lay.off_loss].item())
    rank_ms = [x / steps * 1e3 for x in per_rank]
    # one SUSTAINED figure: >= 2 000 steps in one timed region, no preparation of its own -- what a training run sees
    sustained = None
    if not on_cpu and not args.no_configs and args.workload in ('cfg2', 'cfg4') and not args.lib:
        s_steps = 2000
        s_dt, _ = timed(pool, 0, s_steps)
        sustained = {'steps': s_steps, 'ms_per_step': s_dt / s_steps * 1e3, 'value': n * world * s_steps / s_dt,
                     'what': f'{s_steps} further steps in ONE timed region (HIP events on the launch stream), no warm-up or settle load of its own'}
    # the gradient all-reduce by itself: K back-to-back all-reduces of the flat gradient buffer between two HIP events on the compute
    # stream, MAX over the ranks (what one iteration pays for communication when nothing overlaps it)
    allreduce_us = None
    if dp and not on_cpu:
        for _ in range(5):
            solver._all_reduce(stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if world > 1:
            dist.barrier()
        sync()
        e0.record()
        for _ in range(50):
            solver._all_reduce(stream)
        e1.record()
        sync()
        ar = torch.tensor([e0.elapsed_time(e1) * 1e3 / 50], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        allreduce_us = float(ar.item())
        solver.grads.zero_()

    # north_star's scaling target is STRONG scaling of one 1 048 576-point step of this problem over the GPUs: measured in the
    # same run as a side figure (its own point pool, same step path); the headline stays the per-GPU batch of the workload
    strong = None
    strong_total = 1048576
    if not args.no_strong and not args.global_points and not args.batch and args.workload == 'cfg2' and not on_cpu \
            and strong_total % world == 0:
        ns = strong_total // world
        spool = make_pool(ns)
        s_steps = max(5, min(steps, 20))
        s_dt, _ = timed(spool, 3, s_steps)
        strong = {'global_points': strong_total, 'points_per_gpu': ns, 'steps': s_steps, 'ms_per_step': s_dt / s_steps * 1e3,
                  'value': strong_total * s_steps / s_dt, 'unit': 'points/s', 'scaling': 'strong'}
        del spool

    # matrix-kernel durations, measured live with HIP events around the launches (separate pass, same stream)
    tile_ms = wgrad_ms = None
    kernel = wgrad_kernel = ''
    if rank == 0:
        lib = model.net.lib
        lib.pinn_profile_tile(1)
        s_tile, s_wg = [], []
        for i in range(min(steps, 50)):
            if generic:
                solver._generic_step(pool[i % POOL], ('equation',), [], mse, world)      # (the bracket sits on the backward launch)
            else:
                solver._fused_step(pool[i % POOL], world, stream=stream)
            s_tile.append(float(lib.pinn_last_tile_ms()))
            s_wg.append(float(lib.pinn_last_wgrad_ms()))
        lib.pinn_profile_tile(0)
        tile_ms = float(np.mean(s_tile))
        wgrad_ms = float(np.mean(s_wg)) if min(s_wg) >= 0 else None
        kernel = lib.pinn_last_kernel_name().decode()
        wgrad_kernel = lib.pinn_last_wgrad_kernel_name().decode() if wgrad_ms is not None else ''
        if on_cpu:
            tile_ms = dt / steps * 1e3          # (the emulator has no events)
    if world > 1:
        dist.barrier()

    # the gated split-bf16 GEMM variant (pinn_set_gemm_mode; `--gemm bf16x3` makes it the headline of a run of its own) as a side
    # figure of the default run: the same W + K steps on the same pool, after everything that describes the fp32 kernels
    side_split = None
    if args.gemm == 'fp32' and args.workload in BASELINE_WORKLOADS and not on_cpu and not generic and not args.lib and not args.no_side:
        solver.set_gemm_mode('bf16x3')
        b_dt, _ = timed(pool, warmup, steps)
        if rank == 0:
            side_split = {'gemm': 'bf16x3', 'ms_per_step': b_dt / steps * 1e3, 'value': n * world * steps / b_dt, 'unit': 'points/s',
                          'kernel': model.net.lib.pinn_last_kernel_name().decode(),
                          'dtype': 'f32 operands as 3xbf16 (exact split), six partial products on bf16 MFMA, fp32 accumulate',
                          'what': 'side figure: the same W + K steps with pinn_set_gemm_mode(BF16X3); NOT the headline (`python bench.py --gemm bf16x3` '
                                  'prints its own line with roofline)'}
        solver.set_gemm_mode('fp32')

    comm_info = solver._comm.describe() if (dp and solver._comm is not None) else None
    out = None
    if rank == 0:
        print(f'[bench] gpu: {dt / steps * 1e3:.4f} ms/step by HIP events ({host_dt / steps * 1e3:.4f} by the host clock), '
              f'tile kernel {tile_ms:.4f} ms' + (f', wgrad kernel {wgrad_ms:.4f} ms' if wgrad_ms else ''), file=sys.stderr)
        # algorithmic FLOPs: the reference's formulation (one stream per derivative D(...) asks for). The
        # kernels propagate the second derivatives a Laplacian / wave / heat operator needs as ONE combined stream, so the
        # matrix pipe executes S_exec / S of that figure; both are reported.
        f_pt = flops_per_point(model.layer_dims, spec.n_streams)
        plan = solver.residual_plan
        s_exec = (2 + spec.nd) if (plan is not None and plan.comb_w is not None and not generic) else spec.n_streams
        f_exec = flops_per_point(model.layer_dims, s_exec)
        roof_note = None
        if generic and len(spec.groups) > 1:
            # direction groups (width 512, many directions): kernel_ms is the LAST backward launch -- one group's streams -- so the FLOPs it is
            # priced with are that group's too (the whole-step count over one launch's time read 1.09 of the peak on poisson512)
            s_exec = len(spec.groups[-1][2])
            f_pt = f_exec = flops_per_point(model.layer_dims, s_exec)
            roof_note = ('generic path in %d direction groups: kernel_ms, flops_per_point and frac describe the LAST pinn_jet_backward launch '
                         '(the %d streams of its group); `value` is the whole step' % (len(spec.groups), s_exec))
        kernels_ms = tile_ms + (wgrad_ms or 0.0)
        achieved = f_pt * n / (kernels_ms * 1e-3) / 1e12
        traffic, traffic_src = hbm_traffic(args.workload, kernel, n)
        roof = {'bound': 'mfma', 'kernel': kernel, 'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / PEAK_FP32_MFMA_TFLOPS, 'flops_per_point': f_pt, 'kernel_ms': kernels_ms,
                'executed': {'streams': s_exec, 'flops_per_point': f_exec,
                             'tflops': f_exec * n / (kernels_ms * 1e-3) / 1e12,
                             'frac': f_exec * n / (kernels_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS},
                'traffic': traffic,
                # north_star asks for "HBM GB/s": the fabric-side bytes of the committed PMC passes over this run's kernel time
                'hbm_gbps': (traffic / (kernels_ms * 1e-3) / 1e9) if traffic else None,
                'traffic_source': ((f'{traffic_src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed, same kernel '
                                    'sources (content hash); NOT measured in this run') if traffic else traffic_src) if traffic_src else None}
        if roof_note:
            roof['what'] = roof_note
        if kernel_variant(kernel):
            # split-bf16 kernel: the matrix pipe executes SIX bf16 MFMA flops per fp32-equivalent flop (products a_i b_j, i + j <= 2)
            # and its bound is the dense bf16 MFMA peak; the fp32-equivalent figures stay alongside (they may exceed the fp32
            # MFMA peak: that pipe is not the one in use)
            bf16_tflops = 6.0 * roof['executed']['tflops']
            roof.update({'fp32_equivalent': {'achieved': achieved, 'frac_of_fp32_mfma_peak': achieved / PEAK_FP32_MFMA_TFLOPS,
                                             'executed_tflops': roof['executed']['tflops']},
                         'achieved': bf16_tflops, 'peak': PEAK_BF16_MFMA_TFLOPS, 'frac': bf16_tflops / PEAK_BF16_MFMA_TFLOPS,
                         'what': 'bf16 MFMA flops executed (6 partial products per fp32-equivalent flop of the S-executed GEMMs) '
                                 'over the dense bf16 MFMA peak; fp32_equivalent = SURVEY 8d algorithmic figure'})
        if wgrad_ms:
            # the weight-gradient GEMMs (a third of the hidden-layer FLOPs) run in their own launch for widths >= 128
            inner = sum(a * b for a, b in zip(model.layer_dims[1:-1], model.layer_dims[2:]))
            roof['kernels'] = {
                kernel: {'ms': tile_ms, 'executed_tflops': (f_exec - 2 * s_exec * inner) * n / (tile_ms * 1e-3) / 1e12},
                (wgrad_kernel or 'pinn_wgrad_kernel'): {'ms': wgrad_ms, 'executed_tflops': 2 * s_exec * inner * n / (wgrad_ms * 1e-3) / 1e12}}
        if generic:
            gg = getattr(solver, '_generic_graph', None) or {}
            step_path = ('generic: pinn_jet_forward -> the equation in torch (autograd over its own ops) -> pinn_jet_backward%s -> Adam launch; roofline.kernel_ms covers the backward launch only'
                         % (' [replayed as one launch graph: %d replays]' % gg.get('replays', 0) if gg.get('graph') is not None else ' [eager]'))
        elif not dp:
            step_path = 'fused (Adam inside the reduction launch)'
        else:
            step_path = 'fused step + %s + Adam launch' % (comm_info['all_reduce'] if comm_info else 'no communicator')
        out = {
            'metric': 'collocation-points/sec (residual+grad+Adam step)',
            'value': n * world * steps / dt,
            'unit': 'points/s',
            'n_gpus': world,
            'steps': steps,
            'warmup': warmup,
            'ms_per_step': dt / steps * 1e3,
            'higher_is_better': True,
            'scaling': scaling,
            'vs_baseline': None,
            'dtype': 'f32 operands as 3xbf16 (exact split), six partial products on bf16 MFMA, fp32 accumulate' if kernel_variant(kernel)
                     else ('f32 (weight-gradient GEMMs of the hidden layers: f32 operands as 3xbf16, six partial products, fp32 accumulate)'
                           if wgrad_split(wgrad_kernel) else 'f32'),
            'gemm': args.gemm,
            'tanh': args.tanh,
            'data': f'synthetic (U[low,high)^{model.total} points resident in HBM, PyTorch-default random-init weights, seed 0)',
            'config': {'workload': f"{'BASELINE ' if args.workload in BASELINE_WORKLOADS else ''}{args.workload}: {wl_text}, {n} collocation points per GPU per step, Adam lr 0.005",
                       'points_per_gpu': n, 'global_points': n * world, 'streams': spec.n_streams,
                       'parallelism': f'dp{world}', 'step_path': step_path,
                       'n_ranks_seen': comm_info['n_ranks'] if comm_info else 1,
                       'rank_ms_per_step': {'min': min(rank_ms), 'max': max(rank_ms), 'all': [round(x, 5) for x in rank_ms]},
                       'all_reduce_us': allreduce_us, 'all_reduce_bytes': int(solver.grads.numel()) * 4 if dp else None},
            'final_loss': loss,
            'settled': {'ms_per_step': dt / steps * 1e3, 'value': n * world * steps / dt},
            'cold': cold,
            'strong': strong,
            'gemm_bf16x3': side_split,
            'timing': 'HIP events on the launch stream; host clock %.4f ms/step; `value` / `settled` = steady state: %.2f s of unrelated fp32 GEMM load between the `cold` pass (same W + K steps, taken first) and the warmup steps (clock ramp of a fresh process, --settle)' % (host_dt / steps * 1e3, args.settle),
            'roofline': roof,
        }
        if on_cpu:
            out['data'] = 'PLUMBING TEST on the CPU emulator build -- not a measurement'
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (the other ranks would wait on it)
            out['cpu_baseline'] = cpu_baseline(args.workload, cpu_batch)
        out['sustained'] = sustained
        if (args.workload == 'cfg2' and world == 1 and not on_cpu and not args.no_configs and not args.lib and not args.batch
                and not args.global_points and not args.unfused and args.gemm == 'fp32'):
            # driver time behind ALL five BASELINE configs: compact lines of the other four in the default run
            out['baseline_configs'] = baseline_configs(device, args.no_parity)
        if parity_at_start is not None:
            out['parity_checked'] = parity_at_start
            # the same comparison at the parameters the timed steps left behind: information, not a verdict -- on a (partly) trained net
            # the residual is small and the fp32 reference itself is only good to 1e-4 in the gradients
            # (`ok`: the survey's k = 2 rule per tensor, |ours - f64| <= max(2 |ref32 - f64|, 1e-5 |f64|), like `parity_checked`)
            trained = parity_check(args.workload, solver, generic, mse)
            out['parity_trained_state'] = {k: trained[k] for k in ('ok', 'loss_rel_err', 'worst_gradient_rel_err', 'gradient_rel_err_per_tensor_ours_ref32')}
    if dp:
        solver.end_data_parallel()
        dist.destroy_process_group()
    failed = False
    if rank == 0:
        # a number from a kernel that failed its own check is not a result: no `value`, non-zero exit
        if out.get('parity_checked') is not None and not out['parity_checked']['ok']:
            failed = True
            out['invalid'] = ('parity_checked.ok is false: the step path of this run disagrees with the oracle beyond the rule '
                              '|ours - f64| <= max(2 |ref32 - f64|, 1e-5 |f64|); `value` withheld (measured: %.6g %s)' % (out['value'], out['unit']))
            out['value'] = None
        # the JSON line goes out LAST: RCCL writes its version banner through C stdio, which (piped) is flushed only at
        # exit and would otherwise land behind it
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if failed:
        sys.exit(3)


if __name__ == '__main__':
    main()
