""" Do the kernels depend on the order in which the waves of a workgroup happen to run?  The emulator's PINN_EMU_SHUFFLE=<seed> mode advances
the waves in random order (tests/emu/emu_runtime.cpp): with every barrier in place the gradients are bit-identical to the round-robin run;
built without the tile loop's barriers (-DPINN_ABL=16, the negative control) they are not.
usage: python tools/emu_shuffle_check.py            (product kernels: expect 0.0 everywhere)
       python tools/emu_shuffle_check.py -DPINN_ABL=16      (no barriers: expect large differences) """
import sys, os, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/emu')
import numpy as np, torch
import build_emu
from pydens_amd import engine
import pydens_amd as pa
import pinn_configs as pc
from helpers import load_params
flags = sys.argv[1:]
lib = engine.bind(ctypes.CDLL(build_emu.build(extra_flags=flags, tag='abl16', widths=(64,)) if flags else build_emu.build()))
def grads(name, n, shuffle):
    if shuffle: os.environ['PINN_EMU_SHUFFLE'] = str(shuffle)
    else: os.environ.pop('PINN_EMU_SHUFFLE', None)
    torch.manual_seed(0)
    cfg = pc.make_config(name, pa.D, torch)
    s = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], _lib=lib, device='cpu')
    pts = torch.from_numpy(pc.sample_points(cfg, n, seed=1))
    s._fused_step(pts, 1)
    return s.grads.clone().numpy(), lib.pinn_last_kernel_name().decode()
for name, n in ((('cfg2', 100), ('cfg4', 150)) if flags else (('cfg2', 100), ('cfg4', 150), ('cfg3', 40))):
    g0, k = grads(name, n, 0)
    diffs = []
    for seed in (1, 2, 3):
        g, _ = grads(name, n, seed)
        diffs.append(float(np.abs(g - g0).max()))
    print(name, k, 'max |grad - round-robin grad| over 3 shuffles:', diffs, flush=True)
