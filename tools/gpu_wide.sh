#!/bin/bash
# wide nets (BASELINE configs 3 / 5) in both GEMM modes: bench lines + a correctness spot check.   usage: tools/gpu_wide.sh <tag>
TAG=${1:-wide}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for c in cfg3 cfg5; do for g in fp32 bf16x3; do
  timeout 300 python bench.py --workload $c --gemm $g --no-cpu-baseline --no-strong > $OUT/bench_${c}_$g.txt 2> $OUT/bench_${c}_$g.err; echo "$c $g: $(grep 'bench\] gpu' $OUT/bench_${c}_$g.err)"
done; done
timeout 600 python - <<'PY' 2>&1 | grep -v Warn
import sys, numpy as np, torch
sys.path.insert(0, 'tests')
import pydens_amd as pa, pinn_configs as pc
from helpers import make_solver, export_grads
for name, n in (('cfg3', 262144), ('cfg5', 131072), ('cfg3', 5000)):
    torch.manual_seed(0)
    cfg, solver = make_solver(name, pa)
    pts = torch.from_numpy(pc.sample_points(cfg, n, seed=3)).cuda()
    out = {}
    for mode in ('fp32', 'bf16x3', 'bf16x3'):
        solver.set_gemm_mode(mode); solver._fused_step(pts, 1); torch.cuda.synchronize()
        out.setdefault(mode, []).append([g.astype(np.float64) for g in export_grads(solver)])
    f = lambda x, y: max(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30) for a, b in zip(x, y))
    print(name, n, solver.model.net.lib.pinn_last_wgrad_kernel_name().decode(), 'vs fp32 %.1e repeat %.1e' % (f(out['bf16x3'][0], out['fp32'][0]), f(out['bf16x3'][1], out['bf16x3'][0])))
PY
