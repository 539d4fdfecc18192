#!/usr/bin/env python
""" register / scratch usage of every kernel instantiation of one width's translation unit (compile only, no GPU)
usage: python tools/kres.py <HP> [name filter regex] [-- extra hipcc flags...] """
import re
import subprocess
import sys

hp = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != '--' else ''
extra = sys.argv[sys.argv.index('--') + 1:] if '--' in sys.argv else []
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip', '-Wno-unused-result', '--cuda-device-only',
       '-Rpass-analysis=kernel-resource-usage', '-DPINN_INST_HP=' + hp, *extra, '-c', 'pinn_inst.inc', '-o', '/tmp/kres_py.o']
out = subprocess.run(cmd, cwd='/root/repo/pydens_amd/csrc', capture_output=True, text=True).stderr
rows = []
for block in out.split('Function Name: ')[1:]:
    sym = block.split()[0]
    name = subprocess.run(['c++filt', sym], capture_output=True, text=True).stdout.strip()
    name = name.replace('void ', '').replace('(PinnKArgs)', '')
    get = lambda key: int(re.search(key + r': (\d+)', block).group(1))
    if pat and not re.search(pat, name):
        continue
    rows.append((name, get('VGPRs'), get('AGPRs'), get(r'VGPRs Spill'), get(r'ScratchSize \[bytes/lane\]'), get(r'Occupancy \[waves/SIMD\]')))
for r in sorted(rows):
    print('%-70s vgpr %3d agpr %3d spill %4d scratch %5d B occ %d' % r)
