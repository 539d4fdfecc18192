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
WIDTHS = (16, 32, 64, 128, 256, 512)
ALLACT_WIDTHS = (16, 32, 64, 128, 256)       # (width 512: no second set of full breadth kernels, pydens_amd/csrc/build.py)


def _deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.h', '.inc', '.cpp'))]
    deps += [os.path.join(HERE, f) for f in ('emu_runtime.h', 'emu_runtime.cpp')]
    deps.append(os.path.join(ROOT, 'include', 'pinn.h'))
    deps.append(os.path.join(ROOT, 'tools', 'experiments', 'pinn_chain_kernel.h'))       # (-DPINN_CHAIN=1 builds)
    return deps


def _run(cmd):
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(' '.join(cmd) + '\n' + res.stdout + res.stderr)
    return cmd[-1]


def _fresh(out, deps):
    return os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps)


def build(force=False, extra_flags=(), tag='', widths=(64,)):
    """ extra_flags / tag: a second library with other build knobs (e.g. -DPINN_CHAIN=1) for the kernel instantiations of
    `widths`, kept beside the default one and sharing every other object with it """
    import fcntl
    os.makedirs(BUILD, exist_ok=True)
    with open(os.path.join(BUILD, '.lock'), 'w') as lock:        # several test workers (pytest-xdist) may ask at once: one builds
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build(force, extra_flags, tag, widths)


def _build(force, extra_flags, tag, widths):
    deps = _deps()
    if not (not force and _fresh(OUT, deps)):
        os.makedirs(BUILD, exist_ok=True)
        jobs = [[CXX, *FLAGS, f'-DPINN_INST_HP={hp}', '-c', os.path.join(CSRC, 'pinn_inst.inc'), '-o',
                 os.path.join(BUILD, f'inst_hp{hp}.o')] for hp in WIDTHS]
        jobs += [[CXX, *FLAGS, '-DPINN_INST_HP=64', f'-DPINN_INST_SPLIT={n}', '-c', os.path.join(CSRC, 'pinn_inst.inc'), '-o',
                  os.path.join(BUILD, f'inst_hp64_split{n}.o')] for n in (1, 2)]
        jobs += [[CXX, *FLAGS, f'-DPINN_INST_HP={hp}', '-DPINN_INST_SPLIT=1', '-c', os.path.join(CSRC, 'pinn_inst.inc'), '-o',
                  os.path.join(BUILD, f'inst_hp{hp}_split.o')] for hp in (128, 256)]
        jobs.append([CXX, *FLAGS, '-DPINN_INST_HP=64', '-DPINN_INST_OWN=1', '-c', os.path.join(CSRC, 'pinn_inst.inc'), '-o',
                     os.path.join(BUILD, 'inst_hp64_own1.o')])          # (BASELINE config 2's fp32 kernel: a unit of its own in the product build)
        jobs += [[CXX, *FLAGS, f'-DPINN_INST_HP={hp}', f'-DPINN_INST_ALLACT={part}', '-c', os.path.join(CSRC, 'pinn_inst.inc'), '-o',
                  os.path.join(BUILD, f'inst_hp{hp}_allact{part}.o')] for hp in ALLACT_WIDTHS for part in ((1, 2) if hp >= 128 else (1,))]
        jobs.append([CXX, *FLAGS, '-c', os.path.join(CSRC, 'pinn_abi.cpp'), '-o', os.path.join(BUILD, 'abi.o')])
        jobs.append([CXX, *FLAGS, '-c', os.path.join(HERE, 'emu_runtime.cpp'), '-o', os.path.join(BUILD, 'emu_runtime.o')])
        with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:
            objs = list(pool.map(_run, jobs))
        _run([CXX, '-shared', '-fPIC', *objs, '-o', OUT])
    if not tag:
        return OUT
    tbuild = os.path.join(HERE, '_build_' + tag)
    tout = os.path.join(tbuild, 'libpinn_emu.so')
    if not force and _fresh(tout, deps + [OUT]):
        return tout
    os.makedirs(tbuild, exist_ok=True)
    jobs = [[CXX, *FLAGS, *extra_flags, f'-DPINN_INST_HP={hp}', '-c', os.path.join(CSRC, 'pinn_inst.inc'), '-o',
             os.path.join(tbuild, f'inst_hp{hp}.o')] for hp in widths]
    split_here = 64 in widths                  # the split-bf16 translation units belong to width 64
    if split_here:
        jobs += [[CXX, *FLAGS, *extra_flags, '-DPINN_INST_HP=64', f'-DPINN_INST_SPLIT={n}', '-c', os.path.join(CSRC, 'pinn_inst.inc'),
                  '-o', os.path.join(tbuild, f'inst_hp64_split{n}.o')] for n in (1, 2)]
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:
        objs = list(pool.map(_run, jobs))
    objs += [os.path.join(BUILD, f'inst_hp{hp}.o') for hp in WIDTHS if hp not in widths]
    if not split_here:
        objs += [os.path.join(BUILD, f'inst_hp64_split{n}.o') for n in (1, 2)]
    objs += [os.path.join(BUILD, f'inst_hp{hp}_split.o') for hp in (128, 256)]
    objs.append(os.path.join(BUILD, 'inst_hp64_own1.o'))
    objs += [os.path.join(BUILD, f'inst_hp{hp}_allact{part}.o') for hp in ALLACT_WIDTHS for part in ((1, 2) if hp >= 128 else (1,))]   # (second set of full breadth kernels: the default build's)
    objs += [os.path.join(BUILD, 'abi.o'), os.path.join(BUILD, 'emu_runtime.o')]
    _run([CXX, '-shared', '-fPIC', *objs, '-o', tout])
    return tout


if __name__ == '__main__':
    print(build(force=True))
