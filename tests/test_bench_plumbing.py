""" bench.py must not die on plumbing when the driver calls it for N > 1: `python bench.py --gpus 2` without
a launcher re-executes itself under torch.distributed.run, one rank per device, and rank 0 prints ONE JSON line last. Exercised
here without a GPU: `--device cpu` runs the same script on the emulator build of the library over gloo (plumbing only). """
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, 'emu'))


def _run(extra):
    import build_emu
    lib = build_emu.build()
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'LOCAL_WORLD_SIZE', 'GROUP_RANK')}
    env['OMP_NUM_THREADS'] = '1'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--device', 'cpu', '--lib', lib, '--batch', '128', '--steps', '2',
           '--warmup', '1', '--no-cpu-baseline', '--settle', '0', *extra]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.strip().splitlines() if ln.strip()]
    return json.loads(lines[-1]), res


def test_gpus_2_without_a_launcher_spawns_its_own_ranks():
    line, res = _run(['--gpus', '2'])
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['warmup'] == 1
    assert line['config']['parallelism'] == 'dp2' and line['config']['n_ranks_seen'] == 2
    assert 'torch.distributed.all_reduce' in line['config']['step_path']        # gloo here: the fallback is named in the line
    assert line['config']['global_points'] == 256 and line['value'] > 0
    assert 'launching' in res.stderr                                              # it became its own launcher
    assert sum(1 for ln in res.stdout.splitlines() if ln.startswith('{')) == 1   # ONE JSON line


def test_single_rank_line_carries_the_contract_keys():
    line, _ = _run(['--gpus', '1'])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'settled', 'cold', 'strong'):
        assert key in line, key
    assert line['n_gpus'] == 1 and 'fused' in line['config']['step_path']
