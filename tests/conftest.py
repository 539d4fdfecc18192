import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')
GOLDEN_NAMES = ('cfg1', 'cfg2', 'cfg3', 'cfg4', 'cfg5', 'ode_sigmoid', 'mixed', 'heat3d', 'kdv', 'resnet3')
# round 5 breadth fixtures (nested skips + second-set activations, mixed third order, fourth order): the restatement is pinned on them like
# on the others; the fp64 jet restatement (oracle/jet_f64.py) does not go there -- the KERNELS are checked against them directly
# (tests/test_golden_extras.py; -m gpu twin in test_gpu_parity.py)
GOLDEN_EXTRA = ('nested_acts', 'mixed3', 'biharm', 'act_params', 'mixed31', 'mixed111')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """ the CPU tier (`-m "not gpu"`: ~330 tests, most of them the product kernels on the fiber emulator) takes 16 minutes on one core and
    under 4 on six: where pytest-xdist is installed, no GPU is in sight and nobody asked for a worker count, run it on several workers
    (what `-n 6` does: the options xdist's own hook would set). The GPU tier stays serial -- one device, timing-sensitive tests, and the
    driver records which libraries the test PROCESS loaded. PYDENS_AMD_TEST_WORKERS=0 switches this off, =N picks the count. """
    want = os.environ.get('PYDENS_AMD_TEST_WORKERS', '')
    if want == '0' or not hasattr(config.option, 'numprocesses') or config.option.numprocesses is not None:
        return None                                     # no xdist, switched off, or an explicit -n on the command line
    if getattr(config.option, 'collectonly', False) or getattr(config.option, 'usepdb', False):
        return None
    if hasattr(config, 'workerinput'):
        return None                                     # (an xdist worker itself)
    try:
        import torch
        if torch.cuda.is_available():
            return None
    except ImportError:
        return None
    n = int(want) if want.isdigit() else min(6, os.cpu_count() or 1)
    if n < 2:
        return None
    config.option.numprocesses = n
    config.option.tx = ['popen'] * n
    if getattr(config.option, 'dist', 'no') == 'no':
        config.option.dist = 'load'
    return None


class Golden:
    """ One `tests/golden/<name>.npz` fixture (made by oracle/make_golden.py from the unmodified reference). """
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN_DIR, f'{name}.npz'))
        n = int(z['n_param_tensors'])
        self.params = [z[f'param_{i}'] for i in range(n)]
        self.grads = [None if z[f'grad_{i}'].size == 0 else z[f'grad_{i}'] for i in range(n)]
        self.finals = [z[f'final_{i}'] for i in range(n)]
        self.points = z['points']
        self.u_hat, self.residual, self.loss0 = z['u_hat'], z['residual'], float(z['loss0'])
        self.predict, self.losses, self.lr = z['predict'], z['losses'], float(z['lr'])


@pytest.fixture(params=GOLDEN_NAMES + GOLDEN_EXTRA)
def golden(request):
    return Golden(request.param)


@pytest.fixture(autouse=True)
def _deterministic_torch_rng():
    """ every test starts from the same torch RNG state: nets are initialised from it and `Solver.fit` draws its sampler
    key from it, so without this a test's random initial weights would depend on which tests ran before it """
    import torch
    torch.manual_seed(20240926)
    yield


def params_close(got, want, rtol, atol=3e-7):
    """ trained parameters: relative L2 tolerance plus an absolute floor per element -- a bias that Adam has walked to
    1e-4 carries the fp32 noise of steps of size lr, which no relative tolerance on the bias itself can absorb """
    a, b = np.asarray(got, dtype=np.float64).ravel(), np.asarray(want, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b)) <= rtol * float(np.linalg.norm(b)) + atol * np.sqrt(a.size)


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def pytest_sessionfinish(session, exitstatus):
    """ achieved margins of the sweep tests (tests/helpers.py::record_margin), for profiles/rNN_grad_margins.txt """
    try:
        import helpers
    except ImportError:
        return
    if not helpers.MARGINS:
        return
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    worst = {}
    for test, case, quantity, achieved, bound, arb in helpers.MARGINS:
        key = (test, quantity)
        w = worst.get(key)
        if w is None or achieved / max(bound, 1e-30) > w[1] / max(w[2], 1e-30):
            worst[key] = (case, achieved, bound, arb, (w[4] if w else 0) + 1, (w[5] if w else 0) + int(arb))
        else:
            worst[key] = (w[0], w[1], w[2], w[3], w[4] + 1, w[5] + int(arb))
    with open(os.path.join(out, 'grad_margins.txt'), 'w') as f:
        f.write('# worst achieved margin per (test, quantity): case, achieved relative error, bound, cases, of them arbitrated in fp64\n')
        for (test, quantity), (case, achieved, bound, arb, n, n_arb) in sorted(worst.items()):
            f.write(f'{test:60s} {quantity:10s} worst {achieved:.2e} (bound {bound:.0e}{", fp64 arbiter" if arb else ""}) in {case[:70]}; {n} cases, {n_arb} arbitrated\n')
