""" Structural exclusion of the packed-fp32 hazard of the split-bf16 kernels.

Round 4 caught `v_pk_fma_f32 v[62:63], v[64:65], ...` followed by `v_mov_b32 v64, ...` reading the NEW v64 in lanes 48-63 of its low
half while the SIMD partner wave issued bf16 MFMAs: a write-after-read on a source of a packed fp32 instruction by the instruction
right behind it. hipcc emits such pairs freely (register reuse). For the translation units that hold bf16-MFMA kernels the build
therefore goes through the assembly listing: in every kernel that contains `v_mfma_*_bf16`, no `v_pk_*_f32` may have a source register
overwritten within MIN_DISTANCE issue slots -- `s_nop` is inserted where the compiler's schedule has it closer -- and the patched
listing is verified before it is assembled. (fp32-MFMA kernels are immune: their matrix instructions occupy the vector issue port.)

    hipcc --cuda-device-only -S  ->  patch  ->  clang -x assembler  ->  lld  ->  clang-offload-bundler  ->  hipcc --cuda-host-only
"""
import os
import re
import subprocess

MIN_DISTANCE = 3            # issue slots between a packed fp32 instruction and the first overwrite of one of its sources
BRANCHES = ('s_branch', 's_cbranch', 's_setpc', 's_swappc', 's_call', 's_endpgm', 's_trap')
TWO_DSTS = ('v_swap_b32', 'v_permlane16_swap_b32', 'v_permlane32_swap_b32')      # instructions that write BOTH of their first two operands


def llvm_tool(hipcc, name):
    """ the LLVM tool that belongs to `hipcc` (not a hard-coded /opt/rocm): asked of the compiler driver itself
    (`--print-prog-name`), then $ROCM_PATH, then beside hipcc. Raises FileNotFoundError if there is none. """
    try:
        res = subprocess.run([hipcc, f'--print-prog-name={name}'], capture_output=True, text=True)
        cand = res.stdout.strip().splitlines()[-1] if res.returncode == 0 and res.stdout.strip() else ''
        if cand and os.path.isabs(cand) and os.path.exists(cand):
            return cand
    except OSError:
        pass
    roots = [os.environ.get('ROCM_PATH'), os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), '/opt/rocm']
    for root in roots:
        if root and os.path.exists(os.path.join(root, 'lib', 'llvm', 'bin', name)):
            return os.path.join(root, 'lib', 'llvm', 'bin', name)
    raise FileNotFoundError(f'asm_guard: no `{name}` beside {hipcc} (tried --print-prog-name, $ROCM_PATH, <hipcc>/../lib/llvm/bin, /opt/rocm)')


REG = re.compile(r'v\[(\d+):(\d+)\]|v(\d+)')
NO_VGPR_DST = ('s_', 'global_store', 'scratch_store', 'buffer_store', 'ds_write', 'v_cmp', 'v_mfma', 'ds_swizzle', 'global_atomic', 'ds_add')


def _regs(tok):
    m = REG.fullmatch(tok.strip())
    if not m:
        return None
    if m.group(1) is not None:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def _is_instruction(t):
    return bool(t) and not t.startswith((';', '.', '//')) and not t.endswith(':')


def _slots(t):
    op = t.split()[0]
    return int(t.split()[1], 0) + 1 if op == 's_nop' else 1


def _dst(t):
    parts = t.split(None, 1)
    if len(parts) < 2 or parts[0].startswith(NO_VGPR_DST):
        return None
    ops = [x.strip().split()[0] for x in parts[1].split(',') if x.strip()]
    d = _regs(ops[0]) if ops else None
    if parts[0].startswith(TWO_DSTS) and len(ops) > 1:
        d = (d or set()) | (_regs(ops[1]) or set())
    return d


def _is_branch(t):
    return t.split()[0].startswith(BRANCHES)


def scan_and_patch(lines, patch=True):
    """ lines of a listing -> (new lines, {kernel: [packed instructions, violations found, nops inserted]}) """
    out = list(lines)
    report = {}
    # kernels: from a `_Z...:` label to `.Lfunc_end`
    bounds, start, name = [], None, None
    for i, raw in enumerate(lines):
        t = raw.strip()
        m = re.match(r'^(_Z\S+):', t)
        if m:
            start, name = i, m.group(1)
        elif t.startswith('.Lfunc_end') and start is not None:
            bounds.append((name, start, i))
            start = None
    inserts = {}            # line index -> s_nop operand to insert BEFORE that line
    for name, lo, hi in bounds:
        # instructions AND labels, in listing order: a label (a jump target: whoever arrives there did not come through the lines above
        # it) and a branch (the next instruction executed may be anywhere) end the window a packed instruction is checked in -- ADVICE r5:
        # the linear walk treated them as ordinary one-slot lines, so a packed instruction at the end of a loop body whose source the
        # first instructions of the branch target overwrite went unseen. Rule: a packed instruction must sit MIN_DISTANCE - 1 slots in
        # front of any label or branch behind it (padded with s_nop in front of that line), so that NO successor can be too close.
        idx = [i for i in range(lo + 1, hi)
               if _is_instruction(lines[i].split(';')[0].strip()) or re.match(r'^\.?[A-Za-z_][\w.$]*:$', lines[i].split(';')[0].strip())]
        text = {i: lines[i].split(';')[0].strip() for i in idx}
        if not any(text[i].startswith('v_mfma') and '_bf16' in text[i].split()[0] for i in idx):
            continue
        n_pk = n_bad = n_nop = 0
        for k, i in enumerate(idx):
            t = text[i]
            if t.endswith(':'):
                continue
            op = t.split()[0]
            if not (op.startswith('v_pk_') and op.endswith('_f32')):
                continue
            n_pk += 1
            srcs = set()
            for tok in [x.strip().split()[0] for x in t.split(None, 1)[1].split(',')][1:4]:
                r = _regs(tok)
                if r:
                    srcs |= r
            dist = 0
            for j in idx[k + 1:k + 1 + 2 * MIN_DISTANCE]:
                u = text[j]
                pending = inserts.get(j)
                if pending is not None:
                    dist += pending + 1
                if u.endswith(':') or _is_branch(u):
                    # window boundary: whatever runs next must already be MIN_DISTANCE slots away
                    if dist < MIN_DISTANCE - 1:
                        n_bad += 1
                        if patch:
                            need = MIN_DISTANCE - 1 - dist
                            have = (pending + 1) if pending is not None else 0
                            inserts[j] = have + need - 1
                            n_nop += 1
                    break
                if u.split()[0] in ('s_nop',):
                    dist += _slots(u)
                    continue
                dist += 1
                if dist >= MIN_DISTANCE:
                    break
                d = _dst(u)
                if d and d & srcs:
                    n_bad += 1
                    if patch:
                        need = MIN_DISTANCE - dist                     # slots still missing (an insertion already planned in front of
                        have = (pending + 1) if pending is not None else 0     # the overwriting instruction is counted in `dist`)
                        inserts[j] = have + need - 1
                        n_nop += 1
                    break
        report[name] = [n_pk, n_bad, n_nop]
    if patch and inserts:
        out = []
        for i, raw in enumerate(lines):
            if i in inserts:
                out.append(f'\ts_nop {inserts[i]}\t\t\t; asm_guard: packed-fp32 source overwritten too close behind its reader\n')
            out.append(raw)
    return out, report


def compile_guarded(hipcc, flags, src, obj, verbose=False):
    """ compile `src` (HIP) to the host object `obj` with the device code taken through the guarded listing; returns the report
    {kernel: [packed, violations, nops]} (verified: a second scan of the patched listing finds nothing). Raises on any failure. """
    work = obj + '.guard'
    os.makedirs(work, exist_ok=True)
    s_path, dev_o, dev_out, hipfb = (os.path.join(work, n) for n in ('dev.s', 'dev.o', 'dev.out', 'dev.hipfb'))

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(' '.join(cmd) + '\n' + res.stdout + res.stderr)

    run([hipcc, *flags, '--cuda-device-only', '-S', '-o', s_path, src])
    with open(s_path) as f:
        lines = f.readlines()
    patched, report = scan_and_patch(lines, patch=True)
    _, check = scan_and_patch(patched, patch=False)
    left = {k: v[1] for k, v in check.items() if v[1]}
    if left:
        raise RuntimeError(f'asm_guard: violations left after patching: {left}')
    with open(s_path, 'w') as f:
        f.writelines(patched)
    clang, lld, bundler = (llvm_tool(hipcc, n) for n in ('clang', 'lld', 'clang-offload-bundler'))
    run([clang, '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', s_path, '-o', dev_o])
    run([lld, '-flavor', 'gnu', '-m', 'elf64_amdgpu', '--no-undefined', '-shared', '-o', dev_out, dev_o])
    run([bundler, '-type=o', '-bundle-align=4096',
         '-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950', '-input=/dev/null', f'-input={dev_out}', f'-output={hipfb}'])
    run([hipcc, *flags, '--cuda-host-only', '-Xclang', '-fcuda-include-gpubinary', '-Xclang', hipfb, '-c', src, '-o', obj])
    return report
