""" TEST INFRASTRUCTURE ONLY -- builds tests/emu/_build/libpinn_emu.so: the product's kernel + C-ABI sources
compiled for the HOST against the fiber SIMT emulator (emu_runtime.*), so the kernels' indexing, barriers and
MFMA lane maps can be exercised through the real C-ABI on a machine without a GPU. Never loaded by pydens_amd. """
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'pydens_amd', 'csrc')
BUILD = os.path.join(HERE, '_build')
OUT = os.path.join(BUILD, 'libpinn_emu.so')
CXX = '/opt/rocm/lib/llvm/bin/clang++'
FLAGS = ['-x', 'c++', '-DPINN_EMU', '-O1', '-std=c++17', '-fPIC', '-I', HERE, '-I', CSRC, '-Wno-unknown-pragmas',
         '-Wno-pass-failed']
WIDTHS = (16, 32, 64, 128, 256)


def build(force=False, extra_flags=(), tag=''):
    """ extra_flags / tag: a second library with other build knobs (e.g. -DPINN_CHAIN=1), kept beside the default one """
    global BUILD, OUT
    if tag:
        BUILD = os.path.join(HERE, '_build_' + tag)
        OUT = os.path.join(BUILD, 'libpinn_emu.so')
    else:
        BUILD = os.path.join(HERE, '_build')
        OUT = os.path.join(BUILD, 'libpinn_emu.so')
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.h', '.inc', '.cpp'))]
    deps += [os.path.join(HERE, f) for f in ('emu_runtime.h', 'emu_runtime.cpp')]
    deps.append(os.path.join(ROOT, 'include', 'pinn.h'))
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(BUILD, exist_ok=True)
    jobs = []
    for hp in WIDTHS:
        jobs.append([CXX, *FLAGS, *extra_flags, f'-DPINN_INST_HP={hp}', '-c', os.path.join(CSRC, 'pinn_inst.inc'), '-o',
                     os.path.join(BUILD, f'inst_hp{hp}.o')])
    jobs.append([CXX, *FLAGS, '-c', os.path.join(CSRC, 'pinn_abi.cpp'), '-o', os.path.join(BUILD, 'abi.o')])
    jobs.append([CXX, *FLAGS, '-c', os.path.join(HERE, 'emu_runtime.cpp'), '-o', os.path.join(BUILD, 'emu_runtime.o')])

    def run(cmd):
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(' '.join(cmd) + '\n' + res.stdout + res.stderr)
        return cmd[-1]

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:
        objs = list(pool.map(run, jobs))
    res = subprocess.run([CXX, '-shared', '-fPIC', '-o', OUT, *objs], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(res.stdout + res.stderr)
    return OUT


if __name__ == '__main__':
    print(build(force=True))
