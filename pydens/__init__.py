""" `pydens` -- the reference's import name (pydens/__init__.py:4-5; README.md:26 `from pydens import Solver, NumpySampler`,
tutorial cell 1 `from pydens import Solver, D, V, ConvBlockModel` / `from pydens import NumpySampler as NS`) bound to the
MI355X-native engine: every public name of `pydens_amd` under the name existing scripts and notebooks import, so that they
run untouched. Nothing lives here; the implementation is `pydens_amd` (HIP kernels behind include/pinn.h). """
from pydens_amd import *                                      # noqa: F401,F403  (the NumpySampler family: reference `from batchflow.sampler import *`)
from pydens_amd import Solver, D, V, TorchModel, ConvBlockModel, NumpySampler, NS, Sampler, ConstantSampler, current_model  # noqa: F401

__version__ = '1.0.2'                                         # the reference version whose API this mirrors (pydens/__init__.py:7)
