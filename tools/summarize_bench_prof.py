""" Summary of tools/profile_bench.sh: per-kernel durations (kernel trace) and per-launch counter means of the step's matrix
kernels; writes <dir>/pmc.json in the format bench.py's `roofline.traffic` reads (profiles/r02_<cfg>_pmc.json).
Round 5: a kernel row holds launches of ONE grid size -- the workload's (the commonest grid of that kernel in the run); launches
of another grid (bench.py's 4 096-point parity launches when the profile was taken without --no-parity) are listed apart and
enter no mean, neither durations nor counters (they made AverageNs imply frac 0.992 for cfg5). """
import csv, glob, json, os, re, sys
from collections import defaultdict

cfg, out = sys.argv[1], sys.argv[2]
BATCH = {'cfg2': 65536, 'cfg3': 262144, 'cfg4': 131072, 'cfg5': 131072}.get(cfg, 65536)      # (the breadth workloads of bench.py: 65 536 points)


def short(name):
    name = re.sub(r'^void ', '', name)
    name = name.split('(')[0]
    return name.replace(', ', ',')[:90]


res = {'workload': cfg, 'points_per_launch': BATCH, 'kernels': {}}
MAIN_GRID = {}          # kernel -> Grid_Size_X (threads) of the workload's launches
for path in sorted(glob.glob(os.path.join(out, 'trace', '**', '*kernel_trace.csv'), recursive=True)):
    by_grid = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(path)):
        by_grid[short(row['Kernel_Name'])][int(row.get('Grid_Size_X', 0) or 0)].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
    dur = {}
    for k, grids in by_grid.items():
        main = max(grids, key=lambda g: (len(grids[g]), g))       # the workload's grid: the commonest one (ties: the larger)
        dur[k] = grids[main]
        MAIN_GRID[k] = main
        for g, v in grids.items():
            if g != main and (k.startswith('pinn_tile_kernel') or k.startswith('pinn_wgrad_kernel')):
                print(f'   (left out of the means: {len(v)} launches of {k} with grid {g} instead of {main}, mean {sum(v) / len(v) / 1e3:.2f} us)')
    print('== kernel trace (rocprofv3 --kernel-trace --stats of bench.py --workload %s)' % cfg)
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(f'   {k:90s} calls {len(v):5d}  mean {sum(v) / len(v) / 1e3:10.2f} us  min {min(v) / 1e3:10.2f} us  total {sum(v) / 1e6:9.3f} ms')
        if k.startswith('pinn_tile_kernel') or k.startswith('pinn_wgrad_kernel'):
            res['kernels'].setdefault(k, {})['mean_us'] = sum(v) / len(v) / 1e3
            res['kernels'][k]['calls'] = len(v)
            res['kernels'][k]['grid_threads'] = MAIN_GRID[k]
for sub in ('pmc1', 'pmc2', 'pmc3', 'pmc4'):
    for path in sorted(glob.glob(os.path.join(out, sub, '**', '*counter_collection.csv'), recursive=True)):
        vals = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(path)):
            k = short(row['Kernel_Name'])
            if k in MAIN_GRID and row.get('Grid_Size') and int(row['Grid_Size']) != MAIN_GRID[k]:
                continue                    # a launch of another grid (parity check): not this workload's
            vals[k][row['Counter_Name']].append(float(row['Counter_Value']))
        print(f'== counters ({sub}), per-launch means')
        for k, ctrs in vals.items():
            if not (k.startswith('pinn_tile_kernel') or k.startswith('pinn_wgrad_kernel')):
                continue
            print(f'   {k}')
            for c, v in sorted(ctrs.items()):
                print(f'      {c:30s} {sum(v) / len(v):18.1f}  (n={len(v)})')
                res['kernels'].setdefault(k, {})[c] = sum(v) / len(v)
tile = [k for k in res['kernels'] if k.startswith('pinn_tile_kernel')]
if tile:
    # the dominant kernel by time
    k = max(res['kernels'], key=lambda n: res['kernels'][n].get('mean_us', 0) * res['kernels'][n].get('calls', 0))
    res['kernel'] = tile[0].replace(' ', '')
    fetch = sum(v.get('FETCH_SIZE', 0.0) for v in res['kernels'].values())
    write = sum(v.get('WRITE_SIZE', 0.0) for v in res['kernels'].values())
    res['FETCH_SIZE_KiB'], res['WRITE_SIZE_KiB'] = fetch, write
    res['hbm_bytes_per_launch'] = int(2 * fetch * 1024 + write * 1024)
    res['_comment'] = ('rocprofv3 PMC passes of `bench.py --workload %s` (tools/profile_bench.sh), separate runs per counter '
                       'group; per-launch means. FETCH_SIZE / WRITE_SIZE are KiB summed over the matrix kernels of one step '
                       '(tile kernel + streamed weight-gradient kernel where there is one); on gfx950 FETCH_SIZE counts wide '
                       'coalesced reads at half their bytes (MI355X_MICROARCH.md, HBM section), hence hbm_bytes = '
                       '2*FETCH_SIZE*1024 + WRITE_SIZE*1024. Fabric-side counters: Infinity-Cache hits are included.' % cfg)
    for name, v in res['kernels'].items():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'GRBM_GUI_ACTIVE' in v:
            # MFMA-busy cycles summed over 1024 SIMDs vs GUI-active cycles summed over 8 XCDs
            v['mfma_busy_frac'] = v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (v['GRBM_GUI_ACTIVE'] / 8.0)
            v['effective_clock_GHz'] = v['GRBM_GUI_ACTIVE'] / 8.0 / (v['mean_us'] * 1e3) if v.get('mean_us') else None
            print(f"   {name}: MFMA pipe busy {100 * v['mfma_busy_frac']:.1f} % of SIMD cycles, effective clock "
                  f"{v['effective_clock_GHz']:.2f} GHz" if v['effective_clock_GHz'] else '')
    print('hbm bytes per step (2*FETCH+WRITE): %.1f MB' % (res['hbm_bytes_per_launch'] / 1e6))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydens_amd.csrc import build as hip_build      # noqa: E402
res['kernel_sources_sha1'] = hip_build.kernel_sources_sha1()        # bench.py quotes these bytes only for the same kernel sources
json.dump(res, open(os.path.join(out, 'pmc.json'), 'w'), indent=1)
