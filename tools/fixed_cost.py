""" fixed cost of one launch of a BASELINE tile kernel (prologue: weight staging, x-only pre-pass, first points; epilogue: row sums, partial row) against the cost of a
round of its tile loop: HIP-event kernel times at 2, 4, 8, 16 ... tile rounds per workgroup, straight-line fit.  python tools/fixed_cost.py [cfg2|cfg4] """
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np      # noqa: E402
import kbench           # noqa: E402
from pydens_amd import engine   # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
per_round = 256 * 2 * (16 if cfg == 'cfg2' else 32)          # 256 workgroups x 2 teams x points per tile
rows = []
for rounds in (1, 2, 4, 8, 16, 32):
    n = per_round * rounds
    ms, loss, gsum, _ = kbench.bench(cfg, engine.library_path(), n=n, reps=40, rounds=3)
    t = float(np.median(ms))
    rows.append((rounds, n, t))
    print(f'{cfg}: {rounds:3d} rounds per team  n = {n:8d}  tile kernel {t * 1e3:8.2f} us', flush=True)
x = np.array([r[0] for r in rows], dtype=float)
y = np.array([r[2] for r in rows]) * 1e3
slope, icpt = np.polyfit(x[1:], y[1:], 1)
print(f'{cfg}: kernel time = {icpt:.2f} us fixed + {slope:.3f} us per round  (fit over {int(x[1])} .. {int(x[-1])} rounds); at the BASELINE batch the fixed part is '
      f'{100 * icpt / (icpt + slope * (8 if cfg == "cfg2" else 16)):.1f} % of the launch')
