#!/bin/bash
# GPU side of the round-5 experiment: gpurun_variants/lib_altgz.so (the patch built with -DPINN_ALT_GZ=1, width 128) against the product --
# gradients must be bit-identical (same arithmetic, one barrier fewer per layer), then the same-box A/B
cd /root/repo; export TMPDIR=/tmp; OUT=gpurun_out/altgz; mkdir -p $OUT
python - <<'PY' 2>&1 | grep -v Warn | tee $OUT/identical.txt
import ctypes, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import pinn_configs as pc
import pydens_amd as pa
from pydens_amd import engine
libs = {'product': engine.load_library(), 'altgz': engine.bind(ctypes.CDLL('/root/repo/gpurun_variants/lib_altgz.so'))}
for name, n in (('cfg3', 262144), ('skip128', 65536), ('sin128', 65536)):
    out = {}
    for tag, lib in libs.items():
        torch.manual_seed(0)
        cfg = pc.make_config(name, pa.D, torch)
        s = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], lib=lib)
        xs = torch.from_numpy(pc.sample_points(cfg, n, seed=1)).cuda()
        runs = []
        for _ in range(4):
            s._fused_step(xs, 1); runs.append(s.grads.clone())
        out[tag] = runs
    same = all(torch.equal(a, out['product'][0]) for a in out['altgz'])
    print(name, 'altgz == product, 4 runs each:', same, ' kernel', libs['altgz'].pinn_last_kernel_name().decode(), flush=True)
PY
bash tools/gpu_ab_any.sh altgz "cfg3 skip128 sin128" gpurun_variants/lib_altgz.so | tee $OUT/ab.txt
