""" Round-5 experiment, validated on the emulator only (no GPU minutes were left in round 4): apply tools/experiments/r05_alternating_gz_buffers.patch,
then run this -- gradients of the PINN_ALT_GZ=1 build under shuffled wave scheduling against the product build (expect 0.0 everywhere) and of the
negative control (barrier dropped WITHOUT alternating buffers: expect differences under every seed). Next: same-box A/B on cfg3 / skip128 / sin128
(tools/gpu_ab_any.sh), the barrier behind the data-gradient GEMM is 7.7 % / 4.8 % of those kernels (profiles/r04_wide_chunk_sweep.txt). """
import sys, os, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/emu')
import numpy as np, torch
import build_emu
from pydens_amd import engine
import pydens_amd as pa
import pinn_configs as pc
sys.path.insert(0, '/root/repo/tests')
import test_emu_engine as te
base = engine.bind(ctypes.CDLL(build_emu.build()))
alt = engine.bind(ctypes.CDLL(build_emu.build(extra_flags=['-DPINN_ALT_GZ=1'], tag='altgz', widths=(128,))))
ctl = engine.bind(ctypes.CDLL(build_emu.build(extra_flags=['-DPINN_ALT_GZ=1', '-DPINN_ALT_GZ_SAMEBUF=1'], tag='altgz_ctl', widths=(128,))))
def grads(lib, case, shuffle):
    if shuffle: os.environ['PINN_EMU_SHUFFLE'] = str(shuffle)
    else: os.environ.pop('PINN_EMU_SHUFFLE', None)
    torch.manual_seed(0)
    if case == 'skip':
        eq, kw = te._layout_problems(pa.D, torch, 'burgers', dict(layout='fRa fa f+a R f fa+ fa f', features=[96] * 6 + [1], activation=['Sin', 'SiLU', 'GELU', 'Softplus', 'Tanh']))
        s = pa.Solver(eq, **kw, lib=lib, device='cpu'); pts = torch.from_numpy(np.random.RandomState(1).rand(40, 2).astype(np.float32))
    elif case == 'deep':
        eq, kw = te._layout_problems(pa.D, torch, 'poisson', dict(layout='fa' * 7 + 'f', features=[96] * 7 + [1], activation='Tanh'))
        s = pa.Solver(eq, **kw, lib=lib, device='cpu'); pts = torch.from_numpy(np.random.RandomState(1).rand(40, 2).astype(np.float32))
    else:
        cfg = pc.make_config('cfg3', pa.D, torch)
        s = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], lib=lib, device='cpu'); pts = torch.from_numpy(pc.sample_points(cfg, 40, seed=1))
    s._fused_step(pts, 1)
    return s.grads.clone().numpy()
for case in ('cfg3', 'skip', 'deep'):
    want = grads(base, case, 0)
    for name, lib in (('alternating buffers', alt), ('CONTROL: barrier dropped, same buffer', ctl)):
        out = []
        for seed in (0, 1, 2, 3, 4):
            out.append(float(np.abs(grads(lib, case, seed) - want).max()))
        print(case, name, out, flush=True)
