""" TEST INFRASTRUCTURE ONLY -- stand-in for the un-vendored third-party `batchflow` package.

The reference (`/root/reference/pydens/model_torch.py:12`, `pydens/__init__.py:5`) imports
`batchflow.models.torch.Block` and `batchflow.sampler.*`; `batchflow>=0.8.0` (`pyproject.toml:11`)
is not under /root/reference and is not installable here (no network). This package restates the
two things the PINN hot path needs from it, for layouts made of 'f' and 'a' only, so that the
reference file imports and runs UNMODIFIED inside this container. It is used only by `oracle/`
and `tests/`; the product (`pydens_amd`) never imports it.
"""
