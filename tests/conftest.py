import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')
GOLDEN_NAMES = ('cfg1', 'cfg2', 'cfg3', 'cfg4', 'cfg5', 'ode_sigmoid', 'mixed')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


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


@pytest.fixture(params=GOLDEN_NAMES)
def golden(request):
    return Golden(request.param)


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
