#!/usr/bin/env python
""" Per-phase instruction budget of a tile kernel from its gfx950 listing (hipcc -S): the tile loop (the outermost backward branch
of the kernel) cut at its s_barriers, every instruction put in one class -- what a wave ISSUES per tile, phase by phase.

Classes: mfma | fp (v_fma/mul/add/sub/fmac/..._f32, one lane-op each) | pk (v_pk_*_f32: two lane-ops) | trans (v_exp / v_rcp / ...)
| mov (v_mov, v_accvgpr, v_pk_mov: register shuffles) | sel (v_cndmask, v_cmp, v_bfi: selects / sign transfers) | dpp (row sums,
any VALU with a dpp / row_ modifier) | int (integer / address VALU) | lds | vmem (global / buffer / scratch) | salu | wait (s_waitcnt,
s_nop: slots, not instructions -- s_nop N counts N + 1) | other.

usage: python tools/isa_budget.py file.s '<mangled-name regex>' [--lines]      (or: --build HP [extra flags...] to compile first)
The static count of the loop body is the per-tile count as long as the inner loops are fully unrolled (they are in the
shape-specialised kernels; a leftover inner loop is reported with its trip count unknown). """
import re
import subprocess
import sys
from collections import Counter, OrderedDict

FP = ('v_fma_f32', 'v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_fmac_f32', 'v_fmaak_f32', 'v_fmamk_f32', 'v_mad_f32',
      'v_max_f32', 'v_min_f32', 'v_mul_legacy_f32', 'v_fma_mix', 'v_ldexp_f32', 'v_rndne_f32', 'v_fract_f32', 'v_floor_f32', 'v_trunc_f32',
      'v_cvt_')
TRANS = ('v_exp_f32', 'v_rcp_f32', 'v_log_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_sin_f32', 'v_cos_f32', 'v_rcp_iflag')
MOV = ('v_mov_b32', 'v_mov_b64', 'v_accvgpr', 'v_pk_mov', 'v_swap', 'v_readlane', 'v_writelane', 'v_readfirstlane', 'v_permlane', 'v_perm_b32')
SEL = ('v_cndmask', 'v_cmp', 'v_bfi', 'v_cmpx')


def classify(line):
    op = line.split()[0]
    if op.startswith('v_mfma') or op.startswith('v_smfma'):
        return 'mfma'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_sleep'):
        return 'wait'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
        return 'vmem'
    if op.startswith('v_'):
        if ' row_' in line or 'dpp' in op or ' quad_perm' in line or 'row_bcast' in line or 'row_newbcast' in line:
            return 'dpp'
        if op.startswith('v_pk_') and op.endswith('_f32'):
            return 'pk'
        if op.startswith(TRANS):
            return 'trans'
        if op.startswith(MOV):
            return 'mov'
        if op.startswith(SEL):
            return 'sel'
        if op.startswith(FP):
            return 'fp'
        return 'int'
    return 'other'


def function_body(text, pat):
    lines = text.split('\n')
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\S+):', l)
        if m and re.search(pat, m.group(1)):
            j = i + 1
            while not lines[j].startswith('.Lfunc_end'):
                j += 1
            return m.group(1), lines[i + 1:j]
    raise SystemExit(f'no function matching {pat!r}')


def main():
    args = sys.argv[1:]
    show_lines = '--lines' in args
    args = [a for a in args if a != '--lines']
    path, pat = args[0], args[1]
    sym, body = function_body(open(path).read(), pat)
    name = subprocess.run(['c++filt', sym], capture_output=True, text=True).stdout.strip().replace('void ', '').replace('(PinnKArgs)', '')
    # instructions and labels
    ins = []            # (index in body, text)
    labels = {}
    for i, l in enumerate(body):
        t = l.strip()
        if not t or t.startswith((';', '//')):
            continue
        m = re.match(r'^(\.LBB\S+):', t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if t.startswith('.') or t.endswith(':'):
            continue
        ins.append(t.split(';')[0].strip())
    # backward branches: (target index, branch index); the tile loop = the one spanning the most MFMAs
    loops = []
    for k, t in enumerate(ins):
        m = re.match(r'^s_c?branch\S*\s+(\.LBB\S+)', t)
        if m and m.group(1) in labels and labels[m.group(1)] <= k:
            lo = labels[m.group(1)]
            loops.append((sum(1 for x in ins[lo:k] if x.startswith('v_mfma')), lo, k))
    if not loops:
        raise SystemExit('no loop found')
    loops.sort(reverse=True)
    n_mfma, lo, hi = loops[0]
    inner = [(a, b) for (_, a, b) in loops[1:] if a >= lo and b <= hi and (a, b) != (lo, hi)]
    print(f'{name}\n  tile loop: instructions {lo}..{hi} of {len(ins)} ({hi - lo + 1} in the body, {n_mfma} MFMAs), '
          f'{len(inner)} inner backward branch(es)' + (' -- NOT fully unrolled: counts inside them are per trip' if inner else ''))
    phases = []
    cur = Counter()
    for t in ins[lo:hi + 1]:
        c = classify(t)
        if c == 'barrier':
            phases.append(cur)
            cur = Counter()
            continue
        if c == 'wait' and t.startswith('s_nop'):
            cur['wait'] += int(t.split()[1], 0) + 1
        else:
            cur[c] += 1
    phases.append(cur)
    # the loop wraps: the segment behind the last barrier and the one in front of the first are ONE phase (first layer of the next tile)
    first = phases.pop(0)
    phases[-1] = phases[-1] + first
    cols = ['mfma', 'fp', 'pk', 'trans', 'mov', 'sel', 'dpp', 'int', 'lds', 'vmem', 'salu', 'wait', 'other']
    print('  %-5s' % 'phase' + ''.join('%7s' % c for c in cols) + '   valu   lane-ops')
    tot = Counter()
    for i, p in enumerate(phases):
        valu = sum(p[c] for c in ('fp', 'pk', 'trans', 'mov', 'sel', 'dpp', 'int'))
        lane = p['fp'] + 2 * p['pk'] + p['trans']
        print('  %-5d' % i + ''.join('%7d' % p[c] for c in cols) + '%7d %9d' % (valu, lane))
        tot += p
    valu = sum(tot[c] for c in ('fp', 'pk', 'trans', 'mov', 'sel', 'dpp', 'int'))
    print('  %-5s' % 'sum' + ''.join('%7d' % tot[c] for c in cols) + '%7d %9d' % (valu, tot['fp'] + 2 * tot['pk'] + tot['trans']))
    print(f'  VALU non-MFMA per wave and tile: {valu}; with MFMA: {valu + tot["mfma"]}; arithmetic lane-ops (fp + 2 pk + trans): '
          f'{tot["fp"] + 2 * tot["pk"] + tot["trans"]}; non-arithmetic VALU (mov + sel + dpp + int): {tot["mov"] + tot["sel"] + tot["dpp"] + tot["int"]}')
    if show_lines:
        ops = Counter()
        for t in ins[lo:hi + 1]:
            if classify(t) in ('mov', 'int', 'sel'):
                ops[t.split()[0]] += 1
        print('  non-arithmetic VALU opcodes:', ', '.join(f'{k} {v}' for k, v in ops.most_common(30)))


if __name__ == '__main__':
    main()
