""" Solver.fit rates in the reference's own regime (tutorials: batches of 100 .. 1 500 points, nets of 10 .. 40 units): iterations / s with
the chunk of iterations as ONE launch (pinn_fit_kernel.h, round 5: PYDENS_AMD_FIT_PERSIST=2 one hardware workgroup of virtual workgroups
on one CU -- here for ANY number of sweeps, PYDENS_AMD_FIT_ROUNDS=1000, to see where it stops paying; =1 a grid with a device-scope wait)
and -- =0 -- as launch graphs (round 4). One fresh process per line: python tools/small_fit_rate.py  (runs itself once per case and setting) """
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ('cfg1', 'cfg1_256', 'ode_tanh', 'poisson_10', 'ode_family', 'heat_sigmoid')
if len(sys.argv) == 2:                 # python tools/small_fit_rate.py cfg1,cfg1_nosrc : just these cases
    CASES = tuple(sys.argv[1].split(','))

if len(sys.argv) > 2:
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import pinn_configs as pc
    import pydens_amd as pa
    name, iters = sys.argv[1], int(sys.argv[2])
    if os.environ.get('SMALL_FIT_LIB'):       # an experiment build of the library (tools/variant.sh) instead of the product
        import ctypes
        from pydens_amd import engine
        engine._LIB = engine.bind(ctypes.CDLL(os.environ['SMALL_FIT_LIB']))
    D = pa.D
    torch.manual_seed(0)
    sampler = None
    if name in ('cfg1', 'cfg1_256'):      # BASELINE config 1 (README.md:36-53) at its 100 points; at 256 points (16 tiles: two sweeps of 8)
        cfg = pc.make_config('cfg1', pa.D, torch)
        solver, batch = pa.Solver(cfg['equation'], **cfg['solver_kwargs']), (100 if name == 'cfg1' else 256)
    elif name == 'cfg1_nosrc':            # the same net and batch on an equation WITHOUT an x-only term (no pre-pass program): what the fp64 pre-pass costs the one-CU form
        cfg = pc.make_config('cfg1', pa.D, torch)
        solver, batch = pa.Solver(lambda f, x, y: D(D(f, x), x) + D(D(f, y), y) - 1.0, **cfg['solver_kwargs']), 100
    elif name == 'ode_tanh':              # tutorial cells 12-14
        solver, batch = pa.Solver(lambda f, x: D(f, x) - 2 * np.pi * torch.cos(2 * np.pi * x), ndims=1, initial_condition=.5, activation='Tanh',
                                  layout='fafaf', features=[12, 10, 1]), 400
    elif name == 'poisson_10':            # cells 19-21
        solver, batch = pa.Solver(lambda f, x, y: D(D(f, x), x) + D(D(f, y), y) - 5 * torch.sin(np.pi * (x + y)), ndims=2, boundary_condition=1,
                                  layout='fafaf', features=[10, 10, 1], activation='Tanh'), 400
    elif name == 'ode_family':            # cells 28-31 (default net: 20, 30 units)
        solver, batch = pa.Solver(lambda f, x, e: D(f, x) - e * np.pi * torch.cos(e * np.pi * x), ndims=1, initial_condition=2.0, nparams=1), 700
        sampler = pa.NumpySampler('u') & pa.NumpySampler('u', low=.5, high=5.5)
    else:                                 # cells 37-40 on 30 / 24 units (the notebook's 30 / 40 pads to width 64: outside the one-launch form)
        solver, batch = pa.Solver(lambda f, x, y, t, a: D(D(f, x), x) + D(D(f, y), y) - a * D(f, t), ndims=3, nparams=1,
                                  initial_condition=lambda x, y: 10 * x * y * (1 - x) * (1 - y), boundary_condition=0, layout='fafaf',
                                  features=[30, 24, 1], activation='Sigmoid'), 1500
        sampler = pa.NumpySampler('u', dim=2) & pa.NumpySampler('u', low=0, high=.5) & pa.NumpySampler('u', low=.1, high=4)
    import gc; gc.collect(); gc.disable()          # (as timeit does, in front of the warm-up: tools/fit_one.py)
    solver.fit(niters=300, batch_size=batch, sampler=sampler)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solver.fit(niters=iters, batch_size=batch, sampler=sampler)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    print(f'{name:14s} batch {batch:5d} width {solver.model.net.layout.hp:3d}  {iters / dt:10.0f} it/s  ({dt / iters * 1e6:7.2f} us/it)  '
          f'{solver.model.net.lib.pinn_last_kernel_name().decode()}  loss {float(solver.losses[300]):.4g} -> {float(solver.losses[-1]):.4g}')
    sys.exit(0)

TITLES = {'2': 'the product default: a chunk of up to 128 iterations as ONE launch on ONE CU where the batch is one sweep of its virtual workgroups, launch graphs otherwise',
          '2any': 'chunks as ONE launch on ONE CU for ANY number of sweeps (PYDENS_AMD_FIT_ROUNDS=1000): where the one-CU form stops paying',
          '1': 'chunks as ONE launch of a grid of workgroups (device-scope wait per iteration)',
          '0': 'launch graphs of 128-iteration chunks (round 4)'}
for persist in ('2', '2any', '1', '0'):
    print(f'# PYDENS_AMD_FIT_PERSIST={persist[0]}: ' + TITLES[persist], flush=True)
    env = dict(os.environ, PYDENS_AMD_FIT_PERSIST=persist[0])
    if persist == '2any':
        env['PYDENS_AMD_FIT_ROUNDS'] = '1000'
    for name in CASES:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), name, '12800'], capture_output=True, text=True, env=env)
        lines = [l for l in out.stdout.splitlines() if l.startswith(name)]
        print('\n'.join(lines) if lines else f'{name}: FAILED\n{out.stdout[-300:]}\n{out.stderr[-600:]}', flush=True)
