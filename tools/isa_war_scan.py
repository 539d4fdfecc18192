""" Static scan of a gfx950 assembly listing (hipcc -S) for write-after-read distances between an MFMA's source operands and the next
instruction that writes one of those registers: which instruction, how many issue slots behind the MFMA, which operand (A / B / C).
hipcc's hazard recogniser keeps a minimum distance for SrcC only; this lists what it leaves for SrcA / SrcB.
Usage: python tools/isa_war_scan.py file.s [max_distance] """
import re
import sys
from collections import Counter

REG = re.compile(r'v\[(\d+):(\d+)\]|v(\d+)')


def regs(tok):
    m = REG.fullmatch(tok.strip())
    if not m:
        return None
    if m.group(1) is not None:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def main():
    path = sys.argv[1]
    maxd = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    lines = [l.strip() for l in open(path)]
    ins = [(i, l) for i, l in enumerate(lines) if l and not l.startswith((';', '.', '//')) and not l.endswith(':')]
    found = Counter()
    examples = {}
    for k, (i, l) in enumerate(ins):
        if not l.startswith('v_mfma'):
            continue
        ops = [t.strip() for t in l.split(None, 1)[1].split(',')]
        if len(ops) < 4:
            continue
        srcs = {'A': regs(ops[1]), 'B': regs(ops[2]), 'C': regs(ops[3].split()[0])}
        dist = 0
        for (j, m) in ins[k + 1:k + 1 + 3 * maxd]:
            op = m.split()[0]
            if op.startswith('s_nop'):
                dist += int(m.split()[1]) + 1
                continue
            dist += 1
            if dist > maxd:
                break
            if op.startswith(('s_', 'global_store', 'scratch_store', 'buffer_store', 'ds_write', 'v_cmp', 'v_mfma')):
                continue                    # no VGPR destination (an MFMA's own destination overlapping is the tied-accumulator case)
            dst = regs(m.split(None, 1)[1].split(',')[0]) if len(m.split(None, 1)) > 1 else None
            if not dst:
                continue
            kind = 'load' if op.startswith(('ds_read', 'global_load', 'buffer_load', 'scratch_load')) else ('packed' if op.startswith('v_pk') else 'valu')
            for name, s in srcs.items():
                if s and dst & s:
                    key = (name, kind, dist)
                    found[key] += 1
                    examples.setdefault(key, f'{i + 1}: {l}   ->   {j + 1}: {m}')
    for key in sorted(found):
        print(f'Src{key[0]} overwritten by a {key[1]:6s} instruction {key[2]} issue slot(s) behind the MFMA: {found[key]:4d}   e.g. {examples[key]}')
    if not found:
        print('none')


if __name__ == '__main__':
    main()
