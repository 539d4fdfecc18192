""" TEST INFRASTRUCTURE ONLY -- imports the UNMODIFIED reference file in the build container.

`/root/reference/pydens/model_torch.py` needs `batchflow.models.torch.Block` (model_torch.py:12), which is not
vendored; `oracle/batchflow_shim` provides the stand-in. /root/reference does not exist on the GPU box, so this
loader is only used by `oracle/make_golden.py` and by CPU tests that skip when the reference is absent.
"""
import importlib.util
import os
import sys

REFERENCE_FILE = '/root/reference/pydens/model_torch.py'
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'batchflow_shim')


def reference_available():
    return os.path.isfile(REFERENCE_FILE)


def load_reference():
    """ Returns the reference module (Solver, D, V, ConvBlockModel, TorchModel, current_model). """
    if 'pydens_reference_model_torch' in sys.modules:
        return sys.modules['pydens_reference_model_torch']
    if _SHIM not in sys.path:
        sys.path.insert(0, _SHIM)
    spec = importlib.util.spec_from_file_location('pydens_reference_model_torch', REFERENCE_FILE)
    module = importlib.util.module_from_spec(spec)
    sys.modules['pydens_reference_model_torch'] = module
    spec.loader.exec_module(module)
    return module
