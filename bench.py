""" bench.py -- collocation-points/sec of the full residual + grad + Adam step (BASELINE.json metric).

Workload at every N: BASELINE config 2 (2D Poisson, 4x64 Tanh MLP, batch 65 536 points PER GPU, weak scaling);
a "step" = one fused pinn_residual_step (forward jets, ansatz, residual, MSE, reverse sweep) + gradient
all-reduce over RCCL (N > 1) + pinn_adam_step, on point batches already resident in HBM.

    python bench.py --gpus 1 --steps 100 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0). `roofline`: algorithmic FLOPs of the dominant kernel (pinn_tile_kernel) per launch
= F * points, F = 6*S*sum_{l>=2}(in*out) + 4*d*H1, over its mean launch duration measured with HIP
events on the launch stream (pinn_profile_tile), against the fp32 MFMA peak of MI355X (157.3 TFLOP/s).
`cpu_baseline`: the oracle restatement of the reference step (oracle/pinn_oracle.py, torch CPU ops, all host
cores) timed in the same run on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np      # noqa: E402
import torch            # noqa: E402

import pinn_configs as pc       # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
WORKLOAD = 'cfg2'
BATCH_PER_GPU = 65536
POOL = 8                        # distinct pre-generated point batches cycled through
KERNEL = 'pinn_tile_kernel<64,2,1,2,3,0,true,16>'


def flops_per_point(layer_dims, n_streams):
    d, h1 = layer_dims[0], layer_dims[1]
    inner = sum(a * b for a, b in zip(layer_dims[1:-1], layer_dims[2:]))
    return 6 * n_streams * inner + 4 * d * h1


def cpu_baseline(budget_s=15.0, n_points=16384):
    """ oracle (port of the reference step) on the host cores; bounded sample of the same workload:
    batches of `n_points` points (the reference's throughput is flat in the batch size),
    as many Solver.fit iterations as fit in ~budget_s seconds. The intra-op thread count is tuned first (ATen's CPU
    ops on [N,64] tensors slow down when oversubscribed: 8 threads beat 64 on a 2x64-core host) and reported. """
    from oracle import pinn_oracle as po
    cfg = pc.make_config(WORKLOAD, po.D, torch)
    pts = pc.sample_points(cfg, n_points, seed=0, steps=1)
    ncpu = os.cpu_count() or 1
    best = None
    for threads in sorted({min(t, ncpu) for t in (4, 8, 16, 32)}):
        torch.set_num_threads(threads)
        solver = po.OracleSolver(cfg['equation'], **cfg['solver_kwargs'])
        solver.fit(niters=1, batch_size=n_points, points=pts)            # warm-up
        t0 = time.perf_counter()
        solver.fit(niters=2, batch_size=n_points, points=np.repeat(pts, 2, axis=0))
        per_step = (time.perf_counter() - t0) / 2
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
    return dict(value=n_points * steps / dt, unit='points/s', cores=threads, kind='port',
                sample=f'{steps} Solver.fit iterations of {WORKLOAD} at batch {n_points} '
                       f'(oracle/pinn_oracle.py = reference step restated, torch {torch.__version__} CPU ops, fp32; '
                       f'best of 4/8/16/32 intra-op threads on {ncpu} logical CPUs), {dt:.1f} s')


def hbm_traffic(kernel_name):
    """ HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE,
    separate --pmc runs, gfx950 correction applied: profiles/r01_cfg2_pmc.json says how). None if the committed
    counters are of another kernel. """
    path = os.path.join(ROOT, 'profiles', 'r01_cfg2_pmc.json')
    try:
        with open(path) as f:
            pmc = json.load(f)
    except OSError:
        return None
    if pmc.get('kernel') != kernel_name or pmc.get('points_per_launch') != BATCH_PER_GPU:
        return None
    return pmc['hbm_bytes_per_launch']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--unfused', action='store_true',
                    help='diagnostic: run the N > 1 step path (step, all-reduce, Adam as separate launches) at any N')
    args = ap.parse_args()

    import pydens_amd as pa

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = torch.distributed
    if world > 1 or args.unfused:
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
        dist.init_process_group('nccl', device_id=device)

    torch.manual_seed(0)
    cfg = pc.make_config(WORKLOAD, pa.D, torch)
    solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], device=device)
    assert solver.program is not None, solver.program_error
    model, spec = solver.model, solver.spec
    lay = model.net.layout
    n = args.batch
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    pool = [torch.rand((n, model.total), dtype=torch.float32, device=device, generator=gen) for _ in range(POOL)]

    from pydens_amd.solver import FlatAdam
    solver.optimizer = FlatAdam(model, lr=0.005)
    solver.optimizer.refresh()
    if world > 1:
        dist.broadcast(model.flat, src=0)

    def step(i):
        if world == 1 and not args.unfused:
            solver._fused_step(pool[i % POOL], 1, adam=solver.optimizer)    # Adam fused into the reduction launch
        else:
            solver._fused_step(pool[i % POOL], world)
            dist.all_reduce(solver.grads)
            solver.optimizer.step(solver.grads)

    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss = float(solver.grads[lay.off_loss].item())

    # dominant-kernel duration, measured live with HIP events around the tile-kernel launch (separate pass)
    tile_ms = None
    if rank == 0:
        lib = model.net.lib
        lib.pinn_profile_tile(1)
        samples = []
        for i in range(min(args.steps, 50)):
            solver._fused_step(pool[i % POOL], world)
            samples.append(float(lib.pinn_last_tile_ms()))
        lib.pinn_profile_tile(0)
        tile_ms = float(np.mean(samples))
    if world > 1:
        dist.barrier()

    out = None
    if rank == 0:
        print(f'[bench] gpu: {dt / args.steps * 1e3:.3f} ms/step, tile kernel {tile_ms:.3f} ms', file=sys.stderr)
        # algorithmic FLOPs: the reference's formulation (one stream per derivative D(...) asks for, S = 5 here;
        # SURVEY.md 8d). The kernel itself propagates the two second derivatives as ONE combined stream (S = 4) when the
        # residual allows it, so the matrix pipe executes 4/5 of that figure; both are reported.
        f_pt = flops_per_point(model.layer_dims, spec.n_streams)
        plan = solver.residual_plan
        s_exec = (2 + spec.nd) if (plan is not None and plan.comb_w is not None) else spec.n_streams
        f_exec = flops_per_point(model.layer_dims, s_exec)
        achieved = f_pt * n / (tile_ms * 1e-3) / 1e12
        out = {
            'metric': 'collocation-points/sec (residual+grad+Adam step)',
            'value': n * world * args.steps / dt,
            'unit': 'points/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic (U[0,1)^2 points resident in HBM, PyTorch-default random-init weights, seed 0)',
            'config': {'workload': 'BASELINE cfg2: 2D Poisson u_xx+u_yy=5sin(pi(x+y)), BC=1, 4x64 Tanh MLP, '
                                   f'{n} collocation points per GPU per step, Adam lr 0.005',
                       'points_per_gpu': n, 'global_points': n * world, 'streams': spec.n_streams,
                       'parallelism': f'dp{world}',
                       'step_path': 'fused (Adam inside the reduction launch)' if world == 1 and not args.unfused
                       else 'fused step + all-reduce + Adam launch'},
            'final_loss': loss,
            'roofline': {'bound': 'mfma', 'kernel': KERNEL, 'achieved': achieved,
                         'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP32_MFMA_TFLOPS,
                         'flops_per_point': f_pt, 'kernel_ms': tile_ms,
                         'executed': {'streams': s_exec, 'flops_per_point': f_exec,
                                      'tflops': f_exec * n / (tile_ms * 1e-3) / 1e12,
                                      'frac': f_exec * n / (tile_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS},
                         'traffic': hbm_traffic(KERNEL) if n == BATCH_PER_GPU else None,
                         'traffic_source': 'profiles/r01_cfg2_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)'},
        }
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (the other ranks would wait on it)
            out['cpu_baseline'] = cpu_baseline()
    if world > 1 or args.unfused:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line goes out LAST: RCCL writes its version banner through C stdio, which (piped) is flushed only at
        # exit and would otherwise land behind it
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
