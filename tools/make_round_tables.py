""" markdown tables of a round's bench lines (profiles/rNN_*_bench_line.txt): python tools/make_round_tables.py r05 """
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'


def line(name):
    path = os.path.join(ROOT, 'profiles', f'{tag}_{name}_bench_line.txt')
    return json.loads(open(path).read().strip().splitlines()[-1]) if os.path.exists(path) else None


def busy(name):
    """ MFMA-pipe busy fraction(s) of the matrix kernels from the PMC summary of the same session ('62 / 78 %': tile / weight-gradient kernel) """
    import re
    path = os.path.join(ROOT, 'profiles', f'{tag}_{name}_summary.txt')
    if not os.path.exists(path):
        return ''
    vals = re.findall(r'(pinn_\w+_kernel)<[^>]*>: MFMA pipe busy ([0-9.]+) %', open(path).read())
    return ' / '.join(v for _, v in vals) + ' %' if vals else ''


def traffic(name):
    path = os.path.join(ROOT, 'profiles', f'{tag}_{name}_summary.txt')
    if not os.path.exists(path):
        return ''
    import re
    m = re.search(r'hbm bytes per step \(2\*FETCH\+WRITE\): ([0-9.]+) MB', open(path).read())
    return (f'{float(m.group(1)) / 1e3:.1f} GB' if float(m.group(1)) > 2000 else f'{float(m.group(1)):.0f} MB') if m else ''


print('| workload | ms / step | points/s | algorithmic / executed fraction of 157.3 TF | MFMA pipe busy (tile / wgrad kernel) | L2→fabric bytes / step | bf16x3 ms |')
print('|---|---|---|---|---|---|---|')
names = {'cfg2': 'cfg2 Poisson 4×64, 65 536', 'cfg3': 'cfg3 heat 5×128, 262 144', 'cfg4': 'cfg4 parametric ODE 4×64, 131 072', 'cfg5': 'cfg5 wave 6×256, 131 072'}
for c in ('cfg2', 'cfg3', 'cfg4', 'cfg5'):
    d, s = line(c), line(c + '_split')
    if not d:
        continue
    r = d['roofline']
    tr = r.get('traffic')
    print(f"| {names[c]} | {d['ms_per_step']:.4f} | {d['value']:.3g} | {r['frac']:.3f} / {r['executed']['frac']:.3f} | {busy(c)} | "
          f"{(f'{tr / 1e6:.0f} MB' if tr and tr < 2e9 else (f'{tr / 1e9:.1f} GB' if tr else 'n/a'))} | {s['ms_per_step']:.4f} |" if s else '')
for w in ('skip128', 'skip256', 'sin64', 'sin128', 'gelu256', 'program', 'generic', 'burgers64', 'heat64', 'poisson512'):
    d = line('breadth_' + w)
    if d:
        r = d['roofline']
        print(f"| {w} | {d['ms_per_step']:.3f} | {d['value']:.3g} | {r['frac']:.3f} / {r['executed']['frac']:.3f} | {busy('breadth_' + w)} | {traffic('breadth_' + w)} | parity {d.get('parity_checked', {}).get('ok')} |")
d = line('cfg2_driver_form')
if d:
    print('\ncfg2 driver form (5 + 20 steps):', round(d['ms_per_step'], 4), 'cold', round(d['cold']['ms_per_step'], 4), 'trained-state ok', d['parity_trained_state'].get('ok'),
          'ratios', [round(a / b, 2) for a, b in d['parity_trained_state']['gradient_rel_err_per_tensor_ours_ref32']], 'cpu', d.get('cpu_baseline', {}).get('value'))
