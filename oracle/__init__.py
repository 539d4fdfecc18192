""" TEST INFRASTRUCTURE ONLY.

`oracle/` holds the CPU restatement of the reference's PINN step (pydens/model_torch.py) used to
check the HIP engine. Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import it. The product package `pydens_amd` never imports anything from here and has no CPU
fallback: without its HIP library it raises.

Pinning status: the reference ships NO golden vectors or numeric tests for this path
(pydens/tests/pydens_test.py:13-39 only execs a notebook), so the reference's own tests leave parity
unpinned. We pin the oracle against outputs of the reference file itself, imported unmodified in the
build container with the `batchflow_shim` stand-in (`oracle/make_golden.py` -> `tests/golden/*.npz`).
"""
