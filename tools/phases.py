""" Per-phase cycle breakdown of the tile kernel (needs a -DPINN_PROFILE_PHASES build of the library). """
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pinn_configs as pc
import pydens_amd as pa
from pydens_amd import engine

NAMES = ['0 stage xs+barrier', '1 first layer', '2 barrier', '3 fwd MFMA', '4 fwd epilogue', '5 fwd barrier',
         '6 head dot', '7 barrier', '8 point stage', '9 barrier', '10 bwd act+stage', '11 barrier', '12 wgrad MFMA',
         '13 dgrad MFMA', '14 barrier', '15 layer0 bwd+(5)']
lib_path, cfg_name = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'cfg2')
DUO = len(sys.argv) > 3 and sys.argv[3] == 'duo'
if DUO:
    NAMES = [f'phase {i}' for i in range(14)] + ['idle slot', 'barrier wait']
if len(sys.argv) > 3 and sys.argv[3] == 'chain':      # pinn_chain_kernel
    NAMES = ['0 tile top', '1 first layer', '2 fwd MFMA', '3 fwd activation', '4 head dot', '5 point stage', '6 top reverse',
             '7 act reverse', '8 dgrad MFMA', '9 wgrad stage+MFMA', '10 layer0 reverse', '11 prologue', '12 epilogue', '-', '-', '-']
lib = engine.bind(ctypes.CDLL(lib_path))
lib.pinn_debug_phase_buffer.argtypes = [ctypes.c_void_p]
torch.manual_seed(0)
cfg = pc.make_config(cfg_name, pa.D, torch, V=pa.V)
solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], _lib=lib)
n = min(cfg['n_points'], 131072)
xs = torch.from_numpy(pc.sample_points(cfg, n, seed=1)).cuda()
buf = torch.zeros(1024 * 8 * 16, dtype=torch.int64, device='cuda')
for _ in range(3):
    solver._fused_step(xs, 1)
if len(sys.argv) > 4:
    lib.pinn_debug_set_flags(int(sys.argv[4]))
lib.pinn_debug_phase_buffer(ctypes.c_void_p(buf.data_ptr()))
buf.zero_()
solver._fused_step(xs, 1)
torch.cuda.synchronize()
lib.pinn_debug_phase_buffer(None)
b = buf.cpu().numpy().reshape(-1, 16)
b = b[b.sum(axis=1) > 0]
if len(sys.argv) > 3 and sys.argv[3] == 'chain':
    b = b[b[:, 2] > 0]          # chain waves only (the wgrad waves report their total under 'epilogue')
if len(sys.argv) > 3 and sys.argv[3] == 'teams':
    # two-team kernels (round 6): rows are (workgroup, team, wave) -- the phase totals of team 0 and team 1 side by side
    t = b.reshape(-1, 2, 4, 16)
    print(f'{cfg_name}: {t.shape[0]} workgroups; cycles per wave and launch, team 0 | team 1 (mean over waves and workgroups)')
    for i, name in enumerate(NAMES):
        print(f'  {name:22s} {t[:, 0, :, i].mean():10.0f} | {t[:, 1, :, i].mean():10.0f}')
    print(f'  {"total":22s} {t[:, 0].sum(axis=-1).mean():10.0f} | {t[:, 1].sum(axis=-1).mean():10.0f}')
    sys.exit(0)
nw = b.shape[0]
tot = b.sum(axis=1).mean()
print(f'{cfg_name}: {nw} waves reported, mean total cycles/wave {tot:.0f}')
for i, name in enumerate(NAMES):
    print(f'  {name:22s} {b[:, i].mean():10.0f} cycles/wave  {100 * b[:, i].mean() / tot:5.1f} %   (wave0-of-WG mean {b[0::(8 if DUO else 4), i].mean():9.0f})')
