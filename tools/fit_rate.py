""" User-facing Solver.fit rate (iterations/s, points/s) for the BASELINE configs, default on-device sampler (cfg4: the
README's two-column NumpySampler product). One fresh process per config (tools/fit_one.py): inside one process the
later configs inherit recycled allocator blocks and a warm chip from the earlier ones and cfg3 measured 5 % slower. """
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
for name, iters in (('cfg1', 12800), ('cfg2', 300), ('cfg4', 300), ('cfg3', 40), ('cfg5', 20)):
    out = subprocess.run([sys.executable, os.path.join(HERE, 'fit_one.py'), name, str(iters)], capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith(name)]
    print('\n'.join(lines) if lines else f'{name}: FAILED\n{out.stdout[-400:]}\n{out.stderr[-400:]}', flush=True)
