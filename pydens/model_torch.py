""" `pydens.model_torch` (reference pydens/model_torch.py): the module path some scripts import from. Names only. """
from pydens_amd.tokens import D, V, current_model                # noqa: F401  (:15, :174-188)
from pydens_amd.model import TorchModel, ConvBlockModel          # noqa: F401  (:17-172)
from pydens_amd.solver import Solver                             # noqa: F401  (:191-487)
