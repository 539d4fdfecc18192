""" pydens_amd -- MI355X-native PINN training engine with the user-facing API of `pydens`
(reference pydens/__init__.py:4-5): Solver, D, V, TorchModel, ConvBlockModel and the NumpySampler family.

The hot path of `Solver.fit` (forward MLP, the derivative streams behind `D`, residual loss, reverse sweep, Adam)
runs in hand-written HIP kernels for gfx950 behind the C-ABI of include/pinn.h (libpinn_hip.so). There is no
CPU or eager-PyTorch fallback: constructing a Solver without the built library or without a HIP device raises.
"""
from .tokens import D, V, current_model
from .model import TorchModel, ConvBlockModel
from .solver import Solver
from .sampler import *            # noqa: F401,F403  (reference re-exports batchflow.sampler.*)
from .sampler import NumpySampler, NS, Sampler, ConstantSampler

__version__ = '0.1.0'
