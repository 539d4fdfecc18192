""" Activations of the network (reference model_torch.py:159, :164-168: `getattr(nn, activation)()` for any name, classes, instances,
callables). The kernels evaluate an activation and its derivatives to fourth order from closed forms (pinn_kernel.h pinn_act /
pinn_act_d12 / _d3 / _d4 / pinn_act_zderivs[_ext]); oracle/jet_f64.py states the same formulas in fp64. Here: every formula against
torch's own nested autograd in fp64, and the host side's mapping of what a user may pass to the kernels' activation names.
(The kernels themselves: tests/test_emu_engine.py / test_gpu_parity.py::test_layout_breadth_matches_the_oracle.) """
import numpy as np
import pytest
import torch
from torch import nn

from oracle import jet_f64 as jf

class _Sin(nn.Module):
    def forward(self, x):
        return torch.sin(x)


MODULES = {'sin': _Sin(), 'tanh': nn.Tanh(), 'sigmoid': nn.Sigmoid(), 'softplus': nn.Softplus(), 'silu': nn.SiLU(), 'gelu': nn.GELU(),
           'relu': nn.ReLU(), 'leakyrelu': nn.LeakyReLU(), 'elu': nn.ELU(), 'selu': nn.SELU(), 'softsign': nn.Softsign(),
           'tanhshrink': nn.Tanhshrink(), 'logsigmoid': nn.LogSigmoid(), 'gelu_tanh': nn.GELU(approximate='tanh'), 'mish': nn.Mish(),
           # round 6: module instances configured away from torch's defaults ('name:value', pinn_set_act_params)
           'leakyrelu:0.2': nn.LeakyReLU(0.2), 'elu:0.5': nn.ELU(alpha=0.5), 'softplus:2.0': nn.Softplus(beta=2), 'softplus:0.5': nn.Softplus(beta=0.5),
           'elu:1.7': nn.ELU(alpha=1.7)}


@pytest.mark.parametrize('name', sorted(MODULES))
def test_closed_form_derivatives_equal_nested_autograd(name):
    # (no exact zero in the grid: the piecewise activations have a kink there and autograd's convention is a measure-zero matter)
    z0 = np.concatenate([np.linspace(-6, 6, 40), [0.3, -0.7, 1e-3, -1e-3, 19.0, 25.0, -25.0]])
    z = torch.tensor(z0, dtype=torch.float64, requires_grad=True)
    want = [MODULES[name](z)]
    for _ in range(5):
        g, = torch.autograd.grad(want[-1].sum(), z, create_graph=True, allow_unused=True)
        if g is None or not g.requires_grad:                 # (a linear piece: every further derivative is zero)
            want.append(torch.zeros_like(z) if g is None else g)
            while len(want) < 6:
                want.append(torch.zeros_like(z))
            break
        want.append(g)
    # (value and derivatives 1 .. 4: act_derivs; the fifth -- reverse sweep of fourth-order streams, round 5 -- act_d5)
    got = list(jf.act_derivs(z0, name, fourth=True)) + [jf.act_d5(z0, name)]
    for order, (a, b) in enumerate(zip(got, want)):
        b = b.detach().numpy()
        assert np.abs(np.asarray(a) - b).max() <= 2e-9 * max(1.0, np.abs(b).max()), (name, order)


def test_activation_names_the_host_accepts():
    from pydens_amd.engine import ACT_CODES
    from pydens_amd.model import _activation_name
    F = torch.nn.functional
    for given, want in [('ReLU', 'ReLU'), (nn.ELU, 'ELU'), (nn.LeakyReLU(), 'LeakyReLU'), (nn.GELU(approximate='tanh'), 'GELU_tanh'),
                        (nn.GELU(), 'GELU'), (F.mish, 'Mish'), (torch.relu, 'ReLU'), (F.softsign, 'Softsign'), (nn.SELU(), 'SELU'),
                        (nn.Softplus(), 'Softplus'), (torch.sin, 'Sin'), (F.logsigmoid, 'LogSigmoid'), (nn.Tanhshrink, 'Tanhshrink')]:
        name = _activation_name(given)
        assert name == want and name.lower() in ACT_CODES, (given, name)
    # round 6: LeakyReLU / ELU / Softplus instances carry their one parameter with the name
    for given, want in [(nn.LeakyReLU(0.2), 'LeakyReLU:0.2'), (nn.ELU(alpha=0.5), 'ELU:0.5'), (nn.Softplus(beta=2), 'Softplus:2.0'),
                        (nn.ELU(alpha=1.0), 'ELU')]:
        assert _activation_name(given) == want
    # what the kernels do not implement is refused loudly, not approximated
    for bad in (nn.Softplus(threshold=5), nn.Softplus(beta=-1), nn.PReLU()):
        with pytest.raises(NotImplementedError):
            name = _activation_name(bad)
            if name.lower() not in ACT_CODES:
                raise NotImplementedError(name)
    assert len(set(ACT_CODES.values())) == 16               # every 4-bit activation code of include/pinn.h is taken
