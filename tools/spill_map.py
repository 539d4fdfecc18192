#!/usr/bin/env python
""" Where do the spilled registers of the tile / weight-gradient kernels live?  For every kernel of a width's translation unit
(hipcc -S of pinn_inst.inc): the static spill count (.vgpr_spill_count) and the scratch loads / stores split by where they sit --
in a barrier-delimited segment that holds MFMAs (the GEMM loops: what a spill would cost there is paid per K step) or outside
(prologue, first layer, epilogues, point stage: paid once per tile).
usage: python tools/spill_map.py <HP> [min spills]     (compiles; ~3 min per width) """
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
from pydens_amd.csrc import build      # noqa: E402

hp = int(sys.argv[1])
floor = int(sys.argv[2]) if len(sys.argv) > 2 else 0
asm = f'/tmp/spill_map_{hp}.s'
cmd = ['/opt/rocm/bin/hipcc', *build.FLAGS, *build.WIDTH_FLAGS.get(hp, []), f'-DPINN_INST_HP={hp}', '--cuda-device-only', '-S', '-o', asm,
       os.path.join(HERE, 'pydens_amd', 'csrc', 'pinn_inst.inc')]
if not (os.environ.get('SPILL_MAP_REUSE') and os.path.exists(asm)):
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
text = open(asm).read()
if hp == 64:        # BASELINE config 2's fp32 kernel lives in a unit of its own (pinn_inst.inc PINN_INST_OWN, build.OWN_FLAGS)
    asm2 = f'/tmp/spill_map_{hp}_own1.s'
    cmd2 = ['/opt/rocm/bin/hipcc', *build.FLAGS, *build.OWN_FLAGS[1], f'-DPINN_INST_HP={hp}', '-DPINN_INST_OWN=1', '--cuda-device-only', '-S', '-o', asm2,
            os.path.join(HERE, 'pydens_amd', 'csrc', 'pinn_inst.inc')]
    if not (os.environ.get('SPILL_MAP_REUSE') and os.path.exists(asm2)):
        subprocess.run(cmd2, check=True, stderr=subprocess.DEVNULL)
    text += '\n' + open(asm2).read()
# the second set of full breadth kernels (pinn_inst.inc PINN_INST_ALLACT: 1 = tile kernels, 2 = weight-gradient partners of widths >= 128) lives in
# units of its own: SPILL_MAP_UNITS=allact adds them
if 'allact' in os.environ.get('SPILL_MAP_UNITS', ''):
    for part in ((1, 2) if hp >= 128 else (1,)):
        asm3 = f'/tmp/spill_map_{hp}_allact{part}.s'
        cmd3 = ['/opt/rocm/bin/hipcc', *build.FLAGS, *build.WIDTH_FLAGS.get(hp, []), f'-DPINN_INST_HP={hp}', f'-DPINN_INST_ALLACT={part}', '--cuda-device-only',
                '-S', '-o', asm3, os.path.join(HERE, 'pydens_amd', 'csrc', 'pinn_inst.inc')]
        if not (os.environ.get('SPILL_MAP_REUSE') and os.path.exists(asm3)):
            subprocess.run(cmd3, check=True, stderr=subprocess.DEVNULL)
        text += '\n' + open(asm3).read()
spills = {m.group(1): int(m.group(2)) for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)', text)}
lines = text.split('\n')
rows = []
i = 0
while i < len(lines):
    m = re.match(r'^(_Z\d+pinn_(?:tile|wgrad)_kernel\S+):', lines[i])
    if not m:
        i += 1
        continue
    sym = m.group(1)
    # per barrier-delimited segment: MFMAs, scratch stores / loads, and (round 6) those of them that sit strictly BETWEEN the segment's first
    # and last MFMA -- inside the GEMM's issue span; the others are the epilogue / staging code that shares the segment with a GEMM
    # (a barrier-free loop -- a backward branch -- that holds MFMAs: every scratch instruction of its body is inside the span, wherever it sits in the text)
    def fresh():
        return dict(mfma=0, st=0, ld=0, in_st=0, in_ld=0, pend_st=0, pend_ld=0)
    segs, cur = [], fresh()
    i += 1
    body_start, labels, loops = i, {}, []
    j = i
    while not lines[j].startswith('.Lfunc_end'):
        t = lines[j].strip()
        lm = re.match(r'^(\.LBB\d+_\d+):', t)
        if lm:
            labels[lm.group(1)] = j
        bm = re.match(r'^s_c?branch\S*\s+(\.LBB\d+_\d+)', t)
        if bm and bm.group(1) in labels:
            loops.append((labels[bm.group(1)], j))
        j += 1
    in_mfma_loop = set()
    for lo, hi in loops:
        body = [lines[k].strip() for k in range(lo, hi)]
        # (a loop WITHOUT a barrier inside: a GEMM loop proper. The layer loop of the generic-depth kernels holds whole phases -- barriers, epilogues --
        #  and is read segment by segment like straight-line code)
        if any(t.startswith('v_mfma') for t in body) and not any(t.startswith('s_barrier') for t in body):
            in_mfma_loop.update(range(lo, hi))
    while not lines[i].startswith('.Lfunc_end'):
        t = lines[i].strip()
        if i in in_mfma_loop and t.startswith('scratch_'):
            cur['in_st' if t.startswith('scratch_store') else 'in_ld'] += 1
            cur['st' if t.startswith('scratch_store') else 'ld'] += 1
            i += 1
            continue
        if t.startswith('s_barrier'):
            segs.append(cur)
            cur = fresh()
        elif t.startswith('v_mfma'):
            cur['mfma'] += 1
            cur['in_st'] += cur['pend_st']; cur['in_ld'] += cur['pend_ld']      # (scratch code seen since the previous MFMA lies inside the span)
            cur['pend_st'] = cur['pend_ld'] = 0
        elif t.startswith('scratch_store'):
            cur['st'] += 1
            if cur['mfma']:
                cur['pend_st'] += 1
        elif t.startswith('scratch_load'):
            cur['ld'] += 1
            if cur['mfma']:
                cur['pend_ld'] += 1
        i += 1
    segs.append(cur)
    name = subprocess.run(['c++filt', sym], capture_output=True, text=True).stdout.strip().replace('void ', '').replace('(PinnKArgs)', '')
    gemm = [s for s in segs if s['mfma'] >= 16]
    rest = [s for s in segs if s['mfma'] < 16]
    rows.append((name, spills.get(sym, -1), sum(s['mfma'] for s in gemm), sum(s['in_st'] for s in gemm), sum(s['in_ld'] for s in gemm),
                 sum(s['st'] for s in gemm), sum(s['ld'] for s in gemm), sum(s['st'] for s in rest), sum(s['ld'] for s in rest)))
print(f'width {hp}: kernel | spilled VGPRs | MFMAs | scratch stores / loads BETWEEN the first and last MFMA of a GEMM segment | ... in GEMM segments in all | ... outside')
for r in sorted(rows):
    if r[1] >= floor:
        print('%-66s %4d   mfma %5d   span st %3d ld %3d   gemm-seg st %3d ld %3d   other st %3d ld %3d' % r)
