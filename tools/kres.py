""" register / scratch / LDS usage of the kernels in one translation unit (compile only, no GPU).
usage: python tools/kres.py <file.hip|HP> [symbol regex] [-- extra hipcc flags...]   (HP: pinn_inst.inc at that width) """
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'pydens_amd', 'csrc')
args = sys.argv[1:]
extra = []
if '--' in args:
    i = args.index('--'); args, extra = args[:i], args[i + 1:]
src, pat = args[0], (args[1] if len(args) > 1 else '.')
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip', '-Wno-unused-result',
       '--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', '-I', CSRC, *extra]
if src.isdigit():
    cmd += [f'-DPINN_INST_HP={src}', '-c', os.path.join(CSRC, 'pinn_inst.inc')]
else:
    cmd += ['-c', src]
cmd += ['-o', '/tmp/kres_%d.o' % os.getpid()]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'remark: .*?(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|LDS Size \[bytes/block\]): (\S+)', line)
    if not m:
        if 'error' in line:
            print(line)
        continue
    k, v = m.group(1), m.group(2)
    if k == 'Function Name':
        cur = v; rows[cur] = {}
    elif cur:
        rows[cur][k.split(' ')[0]] = v
for name, r in rows.items():
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace('void ', '').split('(')[0]
    if re.search(pat, dem):
        print(f"{dem:70s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>4} scratch {r.get('ScratchSize','?'):>5} occ {r.get('Occupancy','?')} SGPR {r.get('SGPRs','?')}")
try:
    os.remove('/tmp/kres_%d.o' % os.getpid())
except OSError:
    pass
