""" Builds pydens_amd/libpinn_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

One translation unit per padded hidden width (pinn_inst.inc with -DPINN_INST_HP=...) plus the C-ABI unit,
compiled in parallel, linked into one shared library that sits in-tree next to the Python package so that it
travels to the GPU box with the source snapshot.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, 'libpinn_hip.so')
OBJ = os.path.join(HERE, '_obj')
WIDTHS = (16, 32, 64, 128, 256, 512)
ALLACT_WIDTHS = (16, 32, 64, 128, 256)           # (width 512, round 6: plain kernels + the first set of full breadth kernels only)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip', '-Wno-unused-result', '-I', HERE]
# per-width scheduler choice (same-box A/B of hipcc's -amdgpu-sched-strategy values on the BASELINE kernels, DESIGN.md
# section 6): the width-256 kernels (256 VGPRs, spilling) run 4.9 % faster under `iterative-maxocc` (4.1 % under
# `iterative-ilp`, 10 % slower under `iterative-minreg` / `max-memory-clause`); the other widths are within 1 % of the
# default either way and keep it
# round 2 (two-team width-64 kernels, streamed weight gradients): width 64 gains 1.5 % on cfg4 under `iterative-ilp` (cfg2 +-0);
# width 256 is within 0.3 % between the default and `iterative-maxocc` now (`iterative-ilp` costs its weight-gradient kernel 27 %);
# width 128 is within 1 % everywhere
WIDTH_FLAGS = {256: ['-mllvm', '-amdgpu-sched-strategy=iterative-maxocc'], 64: ['-mllvm', '-amdgpu-sched-strategy=iterative-ilp'],
               512: ['-mllvm', '-amdgpu-sched-strategy=iterative-maxocc']}       # (width 512: register-bound like 256; not measured separately)
# split-bf16 twins of the two BASELINE width-64 kernels (same-box A/B, round 3): 1 = the S = 4 Poisson-box kernel, fastest with the
# SLP vectoriser on (packed fp32 ops: fewer instructions to issue) and -- it sits at the 256-register limit of two waves per SIMD --
# without operand prefetch in its forward / data-gradient GEMMs (40 spilled registers otherwise), 2 = the S = 2 ODE-family kernel,
# fastest without the SLP vectoriser
SPLIT_FLAGS = {1: ['-DPINN_SP_PIPE=0', '-DPINN_SP_PIPE_W=1'], 2: ['-fno-slp-vectorize']}
# (same-box A/B, round 3: width 128 gains 3 % without the SLP vectoriser, width 256 is within 1 % of every flag set tried)
WIDE_SPLIT_FLAGS = {128: ['-fno-slp-vectorize']}
if os.environ.get('PINN_WIDE_SPLIT_FLAGS'):
    import json
    WIDE_SPLIT_FLAGS = {int(k): v for k, v in json.loads(os.environ['PINN_WIDE_SPLIT_FLAGS']).items()}
if os.environ.get('PINN_SPLIT_FLAGS'):
    import json
    SPLIT_FLAGS = {int(k): v for k, v in json.loads(os.environ['PINN_SPLIT_FLAGS']).items()}
if os.environ.get('PINN_SPLIT_EXTRA_FLAGS'):     # experiment builds: JSON list of flags ADDED to both width-64 split units
    import json
    SPLIT_FLAGS = {k: v + json.loads(os.environ['PINN_SPLIT_EXTRA_FLAGS']) for k, v in SPLIT_FLAGS.items()}
OWN_FLAGS = {1: []}                              # pinn_inst.inc PINN_INST_OWN=1: BASELINE config 2's fp32 kernel (default scheduler)
if os.environ.get('PINN_OWN_FLAGS'):             # experiment builds: JSON list of flags for that unit
    import json
    OWN_FLAGS = {1: json.loads(os.environ['PINN_OWN_FLAGS'])}
if os.environ.get('PINN_WIDTH_FLAGS'):          # experiment builds: JSON {width: [flags]} replaces the table above
    import json
    WIDTH_FLAGS = {int(k): v for k, v in json.loads(os.environ['PINN_WIDTH_FLAGS']).items()}


def _sources():
    deps = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(('.h', '.inc', '.cpp'))]
    deps.append(os.path.abspath(__file__))                    # compile flags live here
    deps.append(os.path.join(os.path.dirname(PKG), 'include', 'pinn.h'))
    return deps


def kernel_sources_sha1():
    """ content hash of everything the device code is compiled from (kernel headers, launcher, this file's flags): profiles/*_pmc.json
    carry it, and bench.py only quotes the PMC bytes of a profile taken from the SAME sources """
    import hashlib
    h = hashlib.sha1()
    for name in sorted(f for f in os.listdir(HERE) if f.endswith(('.h', '.inc', '.cpp')) or f == 'build.py'):
        with open(os.path.join(HERE, name), 'rb') as f:
            h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), out=None, obj_dir=None, widths=WIDTHS):
    global OUT, OBJ
    import fcntl
    OUT_, OBJ_ = OUT, OBJ
    if out is not None:
        OUT, OBJ = out, obj_dir or (out + '.obj')
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, '.lock'), 'w') as lock:          # concurrent builds (pytest + a shell) share the objects
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build(force, verbose, extra_flags, widths)
        finally:
            OUT, OBJ = OUT_, OBJ_


def _build(force, verbose, extra_flags, widths):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    deps = _sources()
    if not force and not _stale(OUT, deps):
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for hp in WIDTHS:
        obj = os.path.join(OBJ, f'inst_hp{hp}.o')
        if hp not in widths:
            extra = ['-DPINN_INST_STUB']
        else:
            extra = []
        jobs.append((obj, [hipcc, *FLAGS, *WIDTH_FLAGS.get(hp, []), *extra_flags, *extra, f'-DPINN_INST_HP={hp}', '-c',
                           os.path.join(HERE, 'pinn_inst.inc'), '-o', obj]))
    # second set of full breadth kernels (round 5: all sixteen activations, nested skips -- pinn_inst.inc PINN_INST_ALLACT): a unit of
    # its own per width, so that the build's wall time stays that of its longest unit
    for hp in ALLACT_WIDTHS:
        for part in ((1, 2) if hp >= 128 else (1,)):         # 1: tile kernels (+ the stub of the partner launcher below width 128), 2: weight-gradient partners
            obj = os.path.join(OBJ, f'inst_hp{hp}_allact{part}.o')
            jobs.append((obj, [hipcc, *FLAGS, *WIDTH_FLAGS.get(hp, []), *extra_flags, *([] if hp in widths else ['-DPINN_ONLY_BASELINE']),
                               f'-DPINN_INST_HP={hp}', f'-DPINN_INST_ALLACT={part}', '-c', os.path.join(HERE, 'pinn_inst.inc'), '-o', obj]))
    # split-bf16 kernels of width 64 (pinn_inst.inc, PINN_INST_SPLIT): their own translation units and flags
    for which, flags in SPLIT_FLAGS.items():
        obj = os.path.join(OBJ, f'inst_hp64_split{which}.o')
        jobs.append((obj, [hipcc, *FLAGS, *flags, *extra_flags, '-DPINN_INST_HP=64', f'-DPINN_INST_SPLIT={which}', '-c',
                           os.path.join(HERE, 'pinn_inst.inc'), '-o', obj]))
    # the fp32 kernel of BASELINE config 2: a unit of its own, because it wants the DEFAULT instruction scheduler while the rest of width 64
    # (BASELINE config 4's kernel above all) wants iterative-ilp -- same-box A/B, profiles/r05_headline_ab2.txt (pinn_inst.inc PINN_INST_OWN)
    obj = os.path.join(OBJ, 'inst_hp64_own1.o')
    jobs.append((obj, [hipcc, *FLAGS, *OWN_FLAGS[1], *extra_flags, '-DPINN_INST_HP=64', '-DPINN_INST_OWN=1', '-c',
                       os.path.join(HERE, 'pinn_inst.inc'), '-o', obj]))
    for hp in (128, 256):                 # the split-bf16 WGX tile kernels of BASELINE configs 3 / 5
        obj = os.path.join(OBJ, f'inst_hp{hp}_split.o')
        jobs.append((obj, [hipcc, *FLAGS, *WIDE_SPLIT_FLAGS.get(hp, []), *extra_flags, f'-DPINN_INST_HP={hp}', '-DPINN_INST_SPLIT=1',
                           '-c', os.path.join(HERE, 'pinn_inst.inc'), '-o', obj]))
    obj = os.path.join(OBJ, 'abi.o')
    jobs.append((obj, [hipcc, *FLAGS, *extra_flags, '-c', os.path.join(HERE, 'pinn_abi.cpp'), '-o', obj]))

    guard_report = {}

    def run(job):
        obj, cmd = job
        if not force and not _stale(obj, deps):
            return obj
        if '-DPINN_INST_SPLIT' in ' '.join(cmd) and os.environ.get('PINN_ASM_GUARD', '1') != '0':
            # split-bf16 units: the device code goes through its assembly listing, where no packed fp32 instruction keeps a source
            # that is overwritten within three issue slots (asm_guard.py)
            from pydens_amd.csrc import asm_guard
            src = cmd[cmd.index('-c') + 1]
            flags = [c for c in cmd[1:] if c not in ('-c', src, '-o', obj)]
            try:
                guard_report[os.path.basename(obj)] = asm_guard.compile_guarded(hipcc, flags, src, obj, verbose=verbose)
                return obj
            except FileNotFoundError as err:
                # no assembler / linker / bundler beside this hipcc (a ROCm install laid out differently): the unit is compiled the
                # ordinary way -- LOUDLY: the split-bf16 kernels then lack the structural exclusion of DESIGN.md section 6.2
                # (their two teams stay phase-locked, which is what protected them before round 5)
                import warnings
                warnings.warn(f'pydens_amd build: {err}; compiling {os.path.basename(obj)} WITHOUT the packed-fp32 assembly guard')
                guard_report[os.path.basename(obj)] = {'unguarded': str(err)}
        if verbose:
            print(' '.join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f'hipcc failed for {obj}:\n{res.stdout}\n{res.stderr}')
        if verbose and res.stderr.strip():
            print(res.stderr)
        return obj

    # (the widest units take longest: start them first so that the pool's tail is short)
    order = sorted(range(len(jobs)), key=lambda i: (0 if ('hp256' in jobs[i][0] or 'hp512' in jobs[i][0]) else 1 if 'hp128' in jobs[i][0] else 2, i))
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
        done = dict(zip(order, pool.map(run, [jobs[i] for i in order])))
    objs = [done[i] for i in range(len(jobs))]
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT + '.tmp', *objs]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f'link failed:\n{res.stdout}\n{res.stderr}')
    os.replace(OUT + '.tmp', OUT)
    # what the guard did, beside the library (tests read it; units that were not recompiled keep their earlier entry)
    import json
    rec_path = OUT[:-3] + '.guard.json'
    record = {}
    if os.path.exists(rec_path):
        try:
            with open(rec_path) as f:
                record = json.load(f)
        except (OSError, ValueError):
            record = {}
    record.update(guard_report)
    if os.environ.get('PINN_ASM_GUARD', '1') == '0':
        record = {'disabled': True}
    with open(rec_path, 'w') as f:
        json.dump(record, f, indent=1, sort_keys=True)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
