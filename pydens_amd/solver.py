""" `Solver`: same constructor, `fit`, `predict`, `losses` as the reference (pydens/model_torch.py:191-487); the
per-iteration arithmetic runs in the HIP kernels behind include/pinn.h.

Two step paths share every kernel:
  * fused   -- the equation was lowered to a residual program (trace.py): ONE call of pinn_residual_step does
               forward jets, ansatz, residual, MSE and the whole reverse sweep; then pinn_adam_step. No torch
               arithmetic and no host synchronisation inside the loop (losses stay on the device).
  * generic -- anything else the reference API allows (trainable `V` variables, callable ICs holding variables,
               constraint terms, user criteria): pinn_jet_forward hands the derivative streams to the user's own
               torch code, torch differentiates those few pointwise ops, pinn_jet_backward turns the upstream
               stream gradients into parameter gradients.
Data parallelism: if torch.distributed is initialised, every rank steps on its own `batch_size` points and the flat
gradient buffer (loss slot included) is summed with one all-reduce (RCCL) per iteration.
"""
import contextlib
import math
import os
from contextvars import copy_context

import numpy as np
import torch
from torch import nn
from tqdm import tqdm

from . import engine, trace
from .model import ConvBlockModel, InputProbe, PROBE
from . import tokens
from .tokens import current_model


UNSET = object()        # Solver._imap outside a step: a custom forward()'s input map has not been asked for


class FlatAdam:
    """ torch.optim.Adam semantics (reference model_torch.py:419-422, :461) on the flat kernel buffer. """
    def __init__(self, model, lr=0.005, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kwargs):
        if weight_decay != 0 or amsgrad or kwargs:
            raise NotImplementedError('the HIP Adam kernel implements plain Adam (lr, betas, eps); got '
                                      f'weight_decay={weight_decay}, amsgrad={amsgrad}, {kwargs}')
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        # the moment buffers and the device step counter belong to the MODEL and are zeroed here: every fit call builds a fresh
        # optimizer (reference :419-422) -- at the same device addresses, so that the launch graph a previous fit call recorded
        # (pinn_fit_steps_graph keys on the addresses its nodes carry) is replayed instead of re-recorded. One live
        # FlatAdam per model: the Solver never keeps an old one.
        state = getattr(model, '_adam_state', None)
        if state is None or state[0].shape != model.flat.shape or state[0].device != model.flat.device:
            state = model._adam_state = (torch.zeros_like(model.flat), torch.zeros_like(model.flat),
                                         torch.zeros(1, dtype=torch.int32, device=model.flat.device))
        else:
            for t in state:
                t.zero_()
        self.exp_avg, self.exp_avg_sq, self.step_count = state
        self.t = 0                  # host copy of the step counter (the fused single-rank step passes it by value)
        self.mask = None
        self.members = None         # what this optimizer was built over (reference :420: the parameters that required grad THEN)
        # torch's Adam counts steps PER PARAMETER, and one the loss does not reach is not stepped; the kernel has one count for the whole
        # buffer. Per fit call and trainable scalar (log_scale, V(...) slots): was it reached? -> the step count it has in the reference
        self.calls = []             # (step count at its start, set of unreached offsets) per fit call this optimizer served

    def refresh(self, unreached=()):
        """ unreached: offsets of scalars (log_scale, V(...) slots) the loss terms of the coming fit call do not reach: the reference's backward
        leaves them without a gradient and Adam SKIPS them -- moments, step count and value stay. With fresh moments a zero gradient does the
        same; a reused optimizer (`optimizer=None`) would walk on along the old momentum (Solver._unreached_scalars). """
        new = self.model.trainable_mask()
        if self.members is None:
            # (a buffer of the MODEL, like the mask and the moments: a fresh allocation per fit call shifted what the allocator hands the rest
            #  of the call and cost BASELINE config 4's fit 40 ms per call -- tools/fit_one.py, measured)
            kept = getattr(self.model, '_adam_members', None)
            if kept is None or kept.shape != new.shape or kept.device != new.device:
                self.model._adam_members = kept = torch.empty_like(new)
            kept.copy_(new)
            self.members = kept
        else:
            new = new & self.members        # (`fit(optimizer=None)` after unfreeze_trainable: the reused optimizer never heard of that parameter)
        for off in unreached:
            new[off] = 0
        buf = getattr(self.model, '_adam_mask', None)         # (same address from fit to fit, like the moment buffers)
        if buf is None or buf.shape != new.shape or buf.device != new.device:
            self.model._adam_mask = buf = new
        else:
            buf.copy_(new)
        self.mask = buf
        self.calls.append((self.t, set(unreached)))

    def _served(self):
        """ (iterations, unreached offsets) of every fit call so far; the running one last """
        ends = [t0 for t0, _ in self.calls[1:]] + [self.t]
        return [(end - t0, skipped) for (t0, skipped), end in zip(self.calls, ends)]

    def lagging(self, offsets):
        """ scalars that the coming call reaches but some earlier call of this optimizer did not: their step count in the reference is behind
        the buffer's (other bias corrections) -- what one count for the whole buffer cannot express (Solver._fit hands over to torch's Adam) """
        past = self._served()[:-1]
        return [off for off in offsets if off not in self.calls[-1][1] and any(off in skipped and n > 0 for n, skipped in past)]

    def steps_of(self, off):
        return sum(n for n, skipped in self._served() if off not in skipped)

    def step(self, grads, loss_out=None, stream=None):
        """ loss_out: device address that receives grads[off_loss] in the same launch (the fit's loss history) """
        self.t += 1
        self.model.net.adam_step(self.model.flat, grads, self.exp_avg, self.exp_avg_sq, self.mask, self.step_count,
                                 self.lr, self.betas, self.eps, at=self.t, loss_out=loss_out, stream=stream)


class TorchOptimizerAdapter:
    """ any other `torch.optim` optimizer the reference accepts by name (model_torch.py:420): torch updates the
    parameter views in place; their .grad are views of the flat gradient buffer the kernels fill. """
    def __init__(self, model, name, lr, **kwargs):
        self.model = model
        self.params = model.optimizer_parameters()
        self.opt = getattr(torch.optim, name)(self.params, lr=lr, **kwargs)

    @classmethod
    def continuing(cls, adam):
        """ torch.optim.Adam carrying on where a FlatAdam stands: same members, moments and -- per parameter -- step counts (FlatAdam.lagging) """
        model = adam.model
        self = cls.__new__(cls)
        self.model = model
        members = adam.members
        self.params = [p for p in model.parameters() if bool(members.as_strided(tuple(p.shape), tuple(p.stride()), p.storage_offset()).any())]
        self.opt = torch.optim.Adam(self.params, lr=adam.lr, betas=adam.betas, eps=adam.eps)
        for p in self.params:
            steps = adam.steps_of(p.storage_offset()) if p.numel() == 1 else adam.t
            if steps > 0:
                view = lambda buf: buf.as_strided(tuple(p.shape), tuple(p.stride()), p.storage_offset()).clone()
                self.opt.state[p] = {'step': torch.tensor(float(steps)), 'exp_avg': view(adam.exp_avg), 'exp_avg_sq': view(adam.exp_avg_sq)}
        return self

    def refresh(self):
        pass

    def step(self, grads, loss_out=None, stream=None):
        # Parameters WITHOUT a gradient are skipped by torch's optimizers -- no weight decay, no momentum step, no step count -- and the
        # reference's backward leaves `.grad = None` on (a) a parameter frozen since this optimizer was built (`fit(optimizer=None)` after
        # freeze_trainable) and (b) a scalar the loss of this call does not reach: `log_scale` without an initial condition, a V(...) of the
        # equation in a constraint-only call (round 6's fit-sequence fuzz: AdamW decayed such a variable). The kernels deliver a flat buffer
        # with an exact 0.0 there; a one-entry parameter whose gradient is exactly zero is taken for unreached (one small read-back per step).
        scalars = [p for p in self.params if p.numel() == 1 and p.requires_grad]
        zero = set()
        if scalars:
            values = torch.stack([grads[p.storage_offset()] for p in scalars]).tolist()
            zero = {id(p) for p, v in zip(scalars, values) if v == 0.0}
        for p in self.params:
            reached = p.requires_grad and id(p) not in zero
            p.grad = grads.as_strided(tuple(p.shape), tuple(p.stride()), p.storage_offset()) if reached else None
        self.opt.step()
        for p in self.params:
            p.grad = None


class Solver:
    """ reference model_torch.py:191-487. Extra keyword `device` selects the HIP device (default: current). """
    def __init__(self, equation, ndims, initial_condition=None, boundary_condition=None, domain=(0, 1),
                 nparams=0, model=ConvBlockModel, constraints=None, **kwargs):
        self.equation = equation
        if constraints is None:
            self.constraints = ()
        elif isinstance(constraints, (tuple, list)):
            self.constraints = constraints
        else:
            self.constraints = (constraints, )
        self._losses = []
        self._pending = []
        self.use_fused = True       # diagnostic switch: False keeps every fit on the generic step path
        self.optimizer = None

        self.model = model(**kwargs, ndims=ndims, initial_condition=initial_condition,
                           boundary_condition=boundary_condition, domain=domain, nparams=nparams)
        # the reference's plug-in seam (`Solver(model=...)`, model_torch.py:299-313): subclasses of ConvBlockModel that
        # change how the fully connected net / the ansatz parameters are set up run on the HIP kernels; a subclass that
        # replaces `forward` may put its own torch code AROUND the network -- `self.anzatc(self.conv_block(xs), xs) * g(xs)`, another
        # output transform, no ansatz at all: `self.conv_block(xs)` is then the bare network on the kernels and the rest runs as torch
        # ops on its value and derivative streams (generic step path; `D` applies the chain rule); a fixed map of each point onto as many
        # columns in front of the net is followed too (_input_map). Anything else -- a model that is not a ConvBlockModel, a map in front
        # of the net that is trainable, changes the number of columns or looks at the whole batch -- is refused
        if not isinstance(self.model, ConvBlockModel):
            raise NotImplementedError('Solver(model=...): only ConvBlockModel and its subclasses (fully connected layouts; a custom '
                                      'forward() may wrap torch code around self.conv_block(xs)) are backed by the HIP kernels')
        self.custom_forward = self.model.custom_forward
        self._imap = UNSET                                                # custom forward(): _input_map of the running step
        self._map_pattern, self._map_affine = None, False                 # ... and what _check_input_map found at construction
        current_model.set(self.model)                                     # :316-317
        self.ctx = copy_context()
        self.device = self.model.flat.device
        lay = self.model.net.layout
        self.grads = torch.zeros(lay.p_total, dtype=torch.float32, device=self.device)
        self._generator = None
        self._sample_seed, self._sample_calls = None, 0      # Philox key / batch counter of the device sampler
        self._broadcast_done = False
        self._comm = None                                    # data-parallel communicator (comm.Communicator)

        # "fake run" (:319-325): materialises V-variables and, here, tells which derivative streams D(...) needs
        if self.model.initial_condition is not None and self.model.ic_constant is None and not self.custom_forward:
            fake = torch.rand((3, self.model.total), device=self.device)
            self.ctx.run(self.model.ic_values, fake)
        self._trace_equation()
        # constraint terms (:451-457) the tracer can lower run as further residual programs over the value stream on
        # their own few points; the others keep the generic path (entry None, reason in constraint_errors)
        self.constraint_plans, self.constraint_errors = [], []
        self._born_in_constraint = {}        # constraint number -> variables first created by tracing it (dormant until
        self._constraints_seen = set()       # a fit call has evaluated that constraint: reference :420 vs :457)
        for num, constraint in enumerate(self.constraints):
            before = set(self.model.variables)
            plan, err = self._try_compile_constraint(constraint)
            self.constraint_plans.append(plan)
            self.constraint_errors.append(err)
            self._born_in_constraint[num] = set(self.model.variables) - before
            self.model.dormant_variables |= self._born_in_constraint[num]
        self._traced_constraints = tuple(self.constraints)

    def set_gemm_mode(self, mode):
        """ arithmetic of the hidden-layer GEMMs of the fused step: 'fp32' (default; exact-fp32 MFMA) or 'bf16x3' (every fp32
        operand as three bf16, six partial products, fp32 accumulate -- the split-bf16 kernels where they are built, the fp32
        kernels elsewhere; include/pinn.h pinn_set_gemm_mode). Also settable for a whole process with PYDENS_AMD_GEMM. """
        self.model.net.set_gemm_mode(mode)

    def set_tanh_mode(self, mode):
        """ 'fast' (default) or 'accurate': tanh of small arguments by a minimax polynomial in the kernels that have that form (the
        Poisson-box shape of BASELINE configs 1 / 2 at width 64) -- gradient error on trained models 0.6 - 1.2x the fp32 reference's own
        instead of 1.7 - 1.9x, for +2.5 % kernel time (include/pinn.h pinn_set_tanh_mode). Process-wide: PYDENS_AMD_TANH. """
        self.model.net.set_tanh_mode(mode)

    def _conv_block_argument(self, pts, inside=False):
        """ run the custom forward() up to its call of `self.conv_block` and return that call's argument """
        model = self.model
        model.raw_field = PROBE
        try:
            model.forward(pts) if inside else self.ctx.run(model.forward, pts)         # (inside: the caller runs in self.ctx already)
        except InputProbe as probe:
            ys = probe.arg
        else:
            raise NotImplementedError('Solver(model=...): this forward() never calls self.conv_block(xs) -- a model that is not the '
                                      'fully connected net of its layout is arbitrary torch code the HIP kernels cannot see')
        finally:
            model.raw_field = None
        if not torch.is_tensor(ys):
            raise NotImplementedError('a custom forward() must call self.conv_block on a tensor of points')
        return ys

    def _input_map(self, pts, inside=False):
        """ custom forward(): what does it hand to `self.conv_block`? Asked in every step (a probe call of the forward()).
          None               the batch itself;
          (ys, scale, None)  a fixed per-column affine map `ys[:, k] = a_k * pts[:, k] + b_k` (input normalisation, `2 * xs - 1`): the kernels
                             evaluate the network and its derivative streams AT ys, and a derivative of multi-index alpha with respect to the
                             solver's own columns is the stream times prod_{k in alpha} a_k (_eval_equation);
          (ys, None, pattern) any other fixed map of a point onto as many columns (`torch.sin(xs)`, a time warp, columns mixed): streams at ys
                             with respect to the network's input columns, chain rule with the map's Jacobian by torch autograd
                             (trace.StreamContext._through_map; pattern[k][c] = "ys_k depends on column c", found at construction).
        What the reference's seam (model_torch.py:52-54) also takes and this does not: a map that depends on trainable parameters (its
        gradient would need the streams' own derivatives), on the batch as a whole, or that changes the number of columns (the reference
        builds `conv_block` for `total` inputs, :167): NotImplementedError. """
        pts = pts.detach()
        ys = self._conv_block_argument(pts, inside)
        if (ys.data_ptr() == pts.data_ptr() and ys.shape == pts.shape and ys.stride() == pts.stride() and ys.dtype == pts.dtype
                and ys.device == pts.device):
            return None
        refusal = 'a custom forward() may call self.conv_block only on the batch of points it was given or on a fixed pointwise map of it '
        if ys.requires_grad:
            raise NotImplementedError(refusal + '(this one depends on trainable parameters)')
        if ys.shape != pts.shape:
            raise NotImplementedError(refusal + f'onto as many columns (shape {tuple(ys.shape)} from points of shape {tuple(pts.shape)})')
        ys32 = ys.detach().to(torch.float32).contiguous()
        if self._map_pattern is not None:                   # (known since construction to be more than a per-column affine map)
            return ys32, None, self._map_pattern
        x64, y64 = pts.double(), ys.detach().double()
        if not bool(torch.isfinite(y64).all()):
            raise NotImplementedError(refusal + '(not finite on this batch)')
        lo, hi = x64.argmin(dim=0), x64.argmax(dim=0)
        k = torch.arange(x64.shape[1], device=x64.device)
        span = x64[hi, k] - x64[lo, k]
        flat = span <= 0                                    # a column without two different values: only the identity is recognisable
        a = torch.where(flat, torch.ones_like(span), (y64[hi, k] - y64[lo, k]) / torch.where(flat, torch.ones_like(span), span))
        b = torch.where(flat, torch.zeros_like(span), y64[lo, k] - a * x64[lo, k])     # (such a column must come through unchanged)
        # (fp32 arithmetic of the user's map: a few ulp of its largest intermediate)
        err = (y64 - (a * x64 + b)).abs() - 1e-5 * ((a * x64).abs() + b.abs() + y64.abs())
        if float(err.max()) <= 0.0:
            return ys32, [float(v) for v in a.tolist()], None
        if self._map_affine:
            raise NotImplementedError(refusal + '(per-column affine on the points of the construction, not on this batch: a column without two '
                                      'different values cannot tell its scale)')
        return ys32, None, None

    def _check_input_map(self):
        """ construction: what kind of map does the custom forward() put in front of the network (refusals happen here and not in the first
        fit call), is it a map of each point by itself, and -- if it is not per-column affine -- which input column of the network depends
        on which column of the solver. """
        total = self.model.total
        self._map_pattern, self._map_affine = None, False
        pts = torch.rand((64, total), device=self.device)
        imap = self._input_map(pts)
        if imap is None:
            return
        self._map_affine = imap[1] is not None
        part = self._input_map(pts[:23].contiguous())
        if part is None or not torch.allclose(part[0], imap[0][:23], rtol=1e-6, atol=1e-7):
            raise NotImplementedError('a custom forward() may call self.conv_block only on a map of each point by itself (this one depends on '
                                      'the batch as a whole: the reference differentiates such a map across the rows of the batch, the '
                                      'kernels see one point at a time)')
        if imap[1] is not None:
            return
        p = pts.clone().requires_grad_()
        ys = self._conv_block_argument(p)
        pattern = []
        for k in range(total):
            g = torch.autograd.grad(ys[:, k].sum(), p, retain_graph=True, allow_unused=True)[0] if ys.requires_grad else None
            pattern.append([False] * total if g is None else [bool(v) for v in (g != 0).any(dim=0).tolist()])
        self._map_pattern = pattern

    def _equation_of_the_network(self, net_value, *cols):
        """ custom forward(): the equation as a function of the BARE network's value (tagged: `D` finds its derivative streams)
        and the input columns -- u_hat = model.forward(points) is torch code around it (reference model_torch.py:437-447) """
        model = self.model
        pts = torch.cat(cols, dim=1)                        # = reshape_and_concat of [N,1] columns (:345-362)
        imap = self._imap
        if imap is UNSET:                                   # (the fake run of the tracer: no kernel has run, the question is asked here)
            imap = self._input_map(pts, inside=True)
        if imap is not None and imap[1] is None:
            # a general map in front of the net: the streams are derivatives with respect to the NETWORK's input columns; `D` goes through
            # the map's Jacobian (trace.StreamContext._through_map), taken from the graph this very forward() call builds
            trace.active_streams.get().ymap = {'cols': cols, 'pattern': imap[2], 'ys': None, 'jac': {}}
        model.raw_field = (pts, net_value, None if imap is None else imap[0])
        try:
            u_hat = model.forward(pts)
        finally:
            model.raw_field = None
        return self.equation(u_hat, *cols)

    def _trace_equation(self):
        """ which streams does the equation need, and can it be lowered to a residual program? (construction, and again at
        the start of a fit call whenever the cached lowering no longer reproduces the live callable) """
        self._eq = self._equation_of_the_network if self.custom_forward else self.equation
        if self.custom_forward:
            self._check_input_map()
        self.spec, self.needs_x_grad = trace.discover(self._eq, self.ctx.run, self.model.total, self.device,
                                                      hp=self.model.net.layout.hp, allact=self.model.net.allact)
        if self.custom_forward:
            # torch code between the network and the equation: generic step path only
            self.needs_x_grad = True
            self.ic_var_slot, self.ic_trainable, self.residual_plan = None, False, None
            self.program, self.program_error = None, 'the model has its own forward(): torch code around the network (generic path)'
            self._traced_equation = self.equation
            return
        # a callable initial condition that IS one scalar trainable variable (`lambda *a: V('init', ...)`, reference
        # examples notebook cells 80-88) stays on the fused path: the kernels read it from its user slot and return its
        # gradient there (pinn_residual_t::ic_var1); any other dependence on variables needs torch autograd (generic path)
        self.ic_var_slot = self._ic_variable_slot()
        self.ic_trainable = self.ic_var_slot is None and self._ic_depends_on_variables()
        self.residual_plan = None
        self.program, self.program_error = self._try_compile()
        self._traced_equation = self.equation

    def _refresh_traces(self, nums_constraints):
        """ The reference calls `equation(u_hat, *xs)` and the constraints in EVERY iteration (model_torch.py:448, :457), so
        a closure constant changed between two fit calls, a re-assigned `solver.equation` or new constraint points take
        effect there. Here they were lowered once: re-check the cached lowering against the live callables at the start of
        every fit call (17 random points, a few torch ops) and lower again on any mismatch. """
        if self.equation is not self._traced_equation:
            self._trace_equation()
        elif self.program is not None and self.residual_plan is not None:
            try:
                # (a callable initial condition lowered into the pre-pass is part of the cached lowering: the reference calls
                #  it in every iteration too, model_torch.py:125)
                ok = self._plan_matches(self.residual_plan) and self._ic_matches(self.residual_plan)
            except (trace.TraceUnsupported, NotImplementedError, RuntimeError, TypeError, ValueError):
                ok = False
            if not ok:
                self._trace_equation()
        if tuple(self.constraints) != self._traced_constraints or nums_constraints:
            for num, constraint in enumerate(self.constraints):
                if num < len(self.constraint_plans) and num not in nums_constraints and \
                        num < len(self._traced_constraints) and constraint is self._traced_constraints[num]:
                    continue
                known = num < len(self._traced_constraints) and constraint is self._traced_constraints[num]
                before = set(self.model.variables)
                plan, err = self._try_compile_constraint(constraint)
                if num < len(self.constraint_plans):
                    self.constraint_plans[num], self.constraint_errors[num] = plan, err
                else:
                    self.constraint_plans.append(plan); self.constraint_errors.append(err)
                if not known:
                    # a constraint appended after construction: variables its tracing brings to life are dormant until a fit
                    # call has evaluated it, exactly as for the constraints __init__ saw (reference :420 vs :457)
                    born = set(self.model.variables) - before
                    self._born_in_constraint[num] = self._born_in_constraint.get(num, set()) | born
                    if num not in self._constraints_seen:
                        self.model.dormant_variables |= born
            self._traced_constraints = tuple(self.constraints)

    # ---- tracing ---------------------------------------------------------------------------------------------------
    def _ic_variable_slot(self):
        m = self.model
        if m.initial_condition is None or m.ic_constant is not None:
            return None
        cols = [torch.rand(3, device=self.device) for _ in range(m.ndims_spatial)]
        val = self.ctx.run(m.initial_condition, *cols)
        if isinstance(val, nn.Parameter) and val.numel() == 1 and val.requires_grad:
            return self._variable_slot(val)
        return None

    def _ic_depends_on_variables(self):
        m = self.model
        if m.initial_condition is None or m.ic_constant is not None:
            return False
        fake = torch.rand((3, m.total), device=self.device)
        return bool(self.ctx.run(m.ic_values, fake).requires_grad)

    def _plan_matches(self, plan):
        """ validation on random data: lowered residual (fp64 host interpreter) vs the user's callable on tagged tensors """
        total, n = self.model.total, 17
        streams = torch.rand((self.spec.n_streams, n), device=self.device) * 2 - 1
        pts = torch.rand((n, total), device=self.device) + 0.25
        # (an equation that differentiates composite expressions needs autograd over its own pointwise ops)
        probe = streams.clone().requires_grad_() if self.needs_x_grad else streams
        want = self._eval_equation(probe, pts, requires_grad=self.needs_x_grad)
        want = want.detach().reshape(-1).double().cpu().numpy()
        lay = self.model.net.layout
        var_values = self.model.flat[lay.off_extra:lay.off_extra + plan.n_vars].detach().cpu().numpy()
        got = trace.run_residual_numpy(plan, streams.cpu().numpy(), pts.cpu().numpy(), var_values)
        return bool(np.allclose(got, want, rtol=1e-4, atol=1e-5))

    def _ic_matches(self, plan):
        """ the initial-condition rows of the pre-pass against torch autograd over the live callable on random points """
        if plan.ic_row is None:
            return True
        pts = torch.rand((17, self.model.total), device=self.device) + 0.25
        want = self._ic_stream_tensor(pts, plan.comb_w)
        got = trace.run_ic_numpy(plan, pts.cpu().numpy().astype(np.float64))
        return got.shape == tuple(want.shape) and bool(np.allclose(got, want.double().cpu().numpy(), rtol=1e-4, atol=1e-5))

    def _try_compile(self):
        """ lower the equation to a residual program and cross-check it numerically against the callable. """
        total = self.model.total
        try:
            root = trace.symbolic(self.equation, self.ctx.run, total, variable_slot=self._variable_slot)
            ic_root = self._symbolic_initial_condition()
            if self.spec.mixed3 or self.spec.n4 > 0 or self.spec.mixed111:
                raise trace.TraceUnsupported('mixed third-order partials / fourth-order derivatives run on the generic path')
            plan = trace.lower_residual(root, self.spec, total, ic_root=ic_root)
            trace.combine_second_order(plan, self.spec)
            if not self._plan_matches(plan):
                raise trace.TraceUnsupported('traced program disagrees with the callable (data-dependent control flow?)')
            if plan.comb_w is None and not self.spec.single_call:
                raise trace.TraceUnsupported(f'{self.spec} needs several kernel calls (generic path)')
            self._attach_initial_condition(plan, ic_root)
            self.residual_plan = plan
            return self._with_ic_variable(plan.to_struct()), None
        except trace.TraceUnsupported as err:
            return None, str(err)

    def _symbolic_initial_condition(self):
        """ Sym DAG of a callable IC without trainable variables, or None (constant / trainable / untraceable IC) """
        m = self.model
        self.ic_lowering_error = None
        if m.initial_condition is None or m.ic_constant is not None or self.ic_var_slot is not None or self.ic_trainable:
            return None
        try:
            return trace.symbolic_initial_condition(m.initial_condition, self.ctx.run, m.ndims_spatial)
        except trace.TraceUnsupported as err:
            self.ic_lowering_error = str(err)
            return None

    def _attach_initial_condition(self, plan, ic_root):
        """ callable IC without trainable variables: IC(x) and its derivative streams join the x-only pre-pass (symbolic
        differentiation, trace.attach_initial_condition / lower_residual), so that a fused step holds no torch arithmetic
        at all; anything the tracer cannot lower keeps `_ic_streams` (torch autograd over the callable, per iteration). """
        if ic_root is None:
            return
        try:
            if plan.kind == trace.RES_AFFINE:
                trace.attach_initial_condition(plan, self.spec, ic_root)
            if plan.ic_row is None:
                raise trace.TraceUnsupported('initial condition does not fit the pre-pass')
            # validation against torch autograd over the callable on random points
            if not self._ic_matches(plan):
                raise trace.TraceUnsupported('lowered initial condition disagrees with the callable')
        except (trace.TraceUnsupported, NotImplementedError) as err:
            plan.ic_row, plan.ic_const = None, None          # (unused rows stay in the pre-pass: harmless)
            self.ic_lowering_error = str(err)

    def _try_compile_constraint(self, constraint):
        """ -> ({'program', 'plan', 'points', 'ic'}, None) or (None, reason). """
        model, total = self.model, self.model.total
        if self.custom_forward:
            return None, 'the model has its own forward()'
        if self.ic_trainable:
            return None, 'initial condition holds trainable variables'
        try:
            root, pts = trace.symbolic_constraint(
                constraint, self.ctx.run, total,
                lambda args: self.reshape_and_concat(args).numpy(), variable_slot=self._variable_slot)
            spec0 = trace.StreamSpec(set())
            plan = trace.lower_residual(root, spec0, total)
            # validation: the program (fp64 host interpreter) vs the callable fed with a stand-in for the model's values
            values = torch.rand((pts.shape[0], 1), device=self.device) * 2 - 1
            cols = [torch.from_numpy(pts[:, c:c + 1].copy()).to(self.device) for c in range(total)]
            try:
                want = self.ctx.run(constraint, lambda *args: values, *cols)
            except (TypeError, ValueError, AttributeError, RuntimeError, LookupError) as err:
                raise trace.TraceUnsupported(f'{type(err).__name__}: {err}') from err
            want = torch.as_tensor(want).detach().reshape(-1).double().cpu().numpy()
            lay = model.net.layout
            var_values = model.flat[lay.off_extra:lay.off_extra + plan.n_vars].detach().cpu().numpy()
            got = trace.run_residual_numpy(plan, values.cpu().numpy().reshape(1, -1), pts, var_values)
            if got.shape != want.shape or not np.allclose(got, want, rtol=1e-4, atol=1e-5):
                raise trace.TraceUnsupported('traced constraint disagrees with the callable')
        except trace.TraceUnsupported as err:
            return None, str(err)
        points = torch.from_numpy(pts).to(self.device).contiguous()
        ic = None
        if model.initial_condition is not None and model.ic_constant is None and self.ic_var_slot is None:
            with torch.no_grad():
                ic = self.ctx.run(model.ic_values, points).expand(points.shape[0], 1).reshape(1, -1).float().contiguous()
        return dict(program=self._with_ic_variable(plan.to_struct()), plan=plan, points=points, ic=ic), None

    def _with_ic_variable(self, residual):
        residual.ic_var1 = 0 if self.ic_var_slot is None else self.ic_var_slot + 1
        return residual

    def _constraint_step(self, num, world, accumulate):
        """ gradient + loss of constraint term `num` (mean of its squared values, reference :457), added to / stored in
        `self.grads`; every data-parallel rank evaluates it, hence the 1 / world in front of the all-reduce. """
        cp, model = self.constraint_plans[num], self.model
        n_c = cp['points'].shape[0]
        model.net.residual_step(cp['program'], model.flat, cp['points'], self.grads, model.workspace(n_c, 0, 0), (), 0,
                                ic_streams=cp['ic'], ic_const=model.kernel_ic_const(),
                                inv_n_global=1.0 / (n_c * world), accumulate=accumulate)

    def _variable_slot(self, param):
        """ user slot (index behind `off_extra` in the flat buffer) of a scalar trainable V(...), else None. """
        if param.numel() != 1:
            return None
        lay = self.model.net.layout
        for name, (off, n) in self.model.variables.items():
            if getattr(self.model, name) is param and n == 1:
                return off - lay.off_extra
        return None

    def _eval_equation(self, streams, xs, ic_streams=None, requires_grad=True):
        """ run the user's callable on stream tensors [S,N] (+ optional IC streams), D resolves to streams. """
        total = self.model.total
        sc = trace.StreamContext(total)
        imap = self._imap if self.custom_forward else None
        scale = None if imap is None or imap is UNSET or imap[1] is None or all(a == 1.0 for a in imap[1]) else imap[1]

        def tag(t, alpha):
            # streams taken at an affine map of the points (custom forward(), _input_map): chain rule back to the solver's columns
            if scale is not None and alpha:
                t = t * math.prod(scale[c] for c in alpha)
            return sc.tag(t, alpha)
        full = {}
        for alpha, idx in self.spec.index.items():
            t = streams[idx].view(-1, 1)
            if ic_streams is not None and ic_streams[idx] is not None:
                t = t + ic_streams[idx]
            full[idx] = t
            if all(isinstance(c, int) for c in alpha):          # ('d', a, b) diagonal streams are not user-visible
                # (a node of its own in the autograd graph: the combinations below are built from `full`, NOT from tagged tensors -- `D` of an
                #  expression asks autograd for the partial derivative with respect to every tagged stream, and a mixed partial that hung
                #  below the tagged u_aa would be counted in d / d u_aa a second time)
                tag(t.view_as(t), alpha)
        for ab, (ivv, iaa, ibb) in self.spec.mixed.items():      # u_ab = (u_vv - u_aa - u_bb) / 2
            tag(0.5 * (full[ivv] - full[iaa] - full[ibb]), ab)
        for alpha, (ip, im, ia, ib) in self.spec.mixed4.items():         # u_aabb = (D4_{a+b} + D4_{a-b} - 2 u_aaaa - 2 u_bbbb) / 12
            tag((full[ip] + full[im] - 2.0 * full[ia] - 2.0 * full[ib]) / 12.0, alpha)
        for alpha, terms in list(self.spec.mixed31.items()) + list(self.spec.mixed111.items()):
            # u_aaab / u_abbb from D4 along a +- b and 2a +- b; u_abc from D3 along a +- b +- c (round 6)
            tag(sum(coef * full[idx] for idx, coef in terms), alpha)
        for alpha, (ip, im, i3, sign) in self.spec.mixed3.items():      # u_aab = (D3_{a+b} - D3_{a-b} - 2 u_bbb) / 6, u_abb: + D3_{a-b}, - 2 u_aaa
            tag((full[ip] + sign * full[im] - 2.0 * full[i3]) / 6.0, alpha)
        cols = []
        for c in range(total):
            col = xs[:, c:c + 1]
            if self.needs_x_grad and requires_grad:
                col = col.clone().requires_grad_()
            col._pinn_col = c
            cols.append(col)
        return self.ctx.run(trace.call_with_streams, sc, self._eq, sc.tensors[()], *cols)

    def _ic_streams(self, xs, create_graph):
        """ IC(x_spatial) and its derivative streams as a list over stream indices ([N,1] tensors or None). """
        m, spec = self.model, self.spec
        cols = [xs[:, i].clone().requires_grad_() for i in range(m.ndims_spatial)]
        val = self.ctx.run(m.initial_condition, *cols)
        if not isinstance(val, torch.Tensor):
            val = torch.tensor(float(val), dtype=torch.float32, device=xs.device)
        val = val.to(xs.device).float()
        out = [None] * spec.n_streams
        out[0] = val.view(-1, 1) if val.numel() > 1 else val.reshape(1, 1).expand(xs.shape[0], 1)
        if val.numel() > 1 and val.requires_grad:
            def along(f, direction):
                """ directional derivative of f over the SPATIAL columns of a direction (IC ignores t / parameters) """
                total = None
                for c, weight in trace.dir_weights(direction):
                    if c < m.ndims_spatial and f is not None and f.requires_grad:
                        (g,) = torch.autograd.grad(f.sum(), cols[c], create_graph=True, retain_graph=True, allow_unused=True)
                        if g is not None:
                            g = g if weight == 1.0 else weight * g
                            total = g if total is None else total + g
                return total
            for k, direction in enumerate(spec.dirs):
                g1 = along(val, direction)
                if g1 is None:
                    continue
                out[1 + k] = g1.view(-1, 1)
                if k < spec.n2:
                    g2 = along(g1, direction)
                    if g2 is not None:
                        out[1 + spec.nd + k] = g2.view(-1, 1)
                        if k < spec.n3:
                            g3 = along(g2, direction)
                            if g3 is not None:
                                out[1 + spec.nd + spec.n2 + k] = g3.view(-1, 1)
                                if k < spec.n4:
                                    g4 = along(g3, direction)
                                    if g4 is not None:
                                        out[1 + spec.nd + spec.n2 + spec.n3 + k] = g4.view(-1, 1)
        if not create_graph:
            out = [None if t is None else t.detach() for t in out]
        return out

    def _ic_stream_tensor(self, xs, comb_w):
        """ [S_kernel, N] IC streams by torch autograd over the callable (ICs the tracer could not lower) """
        spec = self.spec
        n2 = spec.n2 if comb_w is None else 1
        parts = self._ic_streams(xs, create_graph=False)
        ic_streams = torch.zeros((1 + spec.nd + n2 + (spec.n3 if comb_w is None else 0), xs.shape[0]), dtype=torch.float32,
                                 device=self.device)
        for i, t in enumerate(parts):
            if t is None:
                continue
            if comb_w is not None and i > spec.nd:
                ic_streams[1 + spec.nd] += comb_w[i - 1 - spec.nd] * t.reshape(-1)
            else:
                ic_streams[i] = t.reshape(-1)
        return ic_streams

    # ---- reference API -----------------------------------------------------------------------------------------------
    @property
    def losses(self):
        """ list of 0-d numpy arrays, one per iteration (reference model_torch.py:308, :464); device-side history is
        fetched lazily so that `fit` itself never synchronises. """
        if self._pending:
            for chunk, done in self._pending:       # `done`: iterations the fit call completed (all, unless interrupted)
                self._losses.extend(np.float32(v) for v in chunk.detach().cpu().numpy().reshape(-1)[:done[0]])
            self._pending = []
            self._losses = [np.asarray(v) for v in self._losses]
        return self._losses

    @losses.setter
    def losses(self, value):
        self._pending, self._losses = [], list(value)

    @classmethod
    def reshape_and_concat(cls, tensors, device=None):
        """ Input casting of reference model_torch.py:327-362 -> float32 [N, D] (on `device` if given). """
        items = list(tensors)
        lengths = [int(np.prod(np.shape(t))) for t in items
                   if isinstance(t, (np.ndarray, torch.Tensor, tuple, list))]
        n = max(lengths) if lengths else 1
        cols = []
        for x in items:
            if isinstance(x, torch.Tensor):
                col = x.reshape(-1, 1).float()
            elif isinstance(x, np.ndarray):
                arr = x if x.size == n else np.tile(x.squeeze()[0], (n, 1))            # :355-356
                col = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32).reshape(n, 1))
            elif isinstance(x, (list, tuple)):
                col = torch.tensor(x, dtype=torch.float32).view(-1, 1)
            else:
                col = torch.full((n, 1), float(x), dtype=torch.float32)
            if device is not None:
                col = col.to(device)
            cols.append(col)
        return torch.cat(cols, dim=1)

    def _group_rows(self, num, idx):
        """ stream rows of direction group `num` as a device index tensor, built once (indexing with a Python list uploads it in every
        step, which a launch-graph recording refuses) """
        cache = self.__dict__.setdefault('_group_row_cache', {})
        key = (num, tuple(idx))
        if key not in cache:
            cache[key] = torch.tensor(list(idx), dtype=torch.long, device=self.device)
        return cache[key]

    def _points_on_device(self, pts):
        """ reshape_and_concat(pts) on the device. Small HOST constants -- the fixed points a constraint builds in every call,
        `f(torch.tensor([0.5]))` -- are looked up by content in a cache of device tensors: no host-to-device copy per iteration, and
        nothing in the step that a launch-graph recording would refuse (pageable-memory copies are not capturable). """
        key = []
        for x in pts:
            if isinstance(x, torch.Tensor):
                if x.is_cuda or x.requires_grad or x.numel() > 4096:
                    return self.reshape_and_concat(pts, device=self.device)
                try:
                    key.append(('t', str(x.dtype), tuple(x.shape), x.detach().contiguous().numpy().tobytes()))
                except (TypeError, RuntimeError):       # (a dtype numpy does not have: no cache entry, the plain upload)
                    return self.reshape_and_concat(pts, device=self.device)
            elif isinstance(x, np.ndarray):
                if x.size > 4096:
                    return self.reshape_and_concat(pts, device=self.device)
                key.append(('a', str(x.dtype), x.shape, np.ascontiguousarray(x).tobytes()))
            elif isinstance(x, (list, tuple, int, float)):
                key.append(('v', repr(x)))
            else:
                return self.reshape_and_concat(pts, device=self.device)
        cache = self.__dict__.setdefault('_host_constants', {})
        key = tuple(key)
        if key not in cache:
            if len(cache) >= 64:
                cache.clear()
            cache[key] = self.reshape_and_concat(pts, device=self.device)
        return cache[key]

    def _world(self):
        dist = torch.distributed
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def _device_guard(self):
        """ the library launches on the CURRENT device: make the model's device current where it is not """
        if self.device.type == 'cuda' and torch.cuda.current_device() != self.device.index:
            return torch.cuda.device(self.device)
        return contextlib.nullcontext()

    def _new_sample_seed(self):
        """ Philox key of the device sampler: drawn from torch's default generator, like the `torch.rand` calls of the
        reference's default sampler (model_torch.py:431) follow `torch.manual_seed`; per-rank offset under data parallelism """
        rank, _ = self._world()
        base = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        self._sample_seed = (base + 7919 * rank) & (2 ** 64 - 1)

    def _sample(self, batch_size, sampler, stream=None):
        """ reference model_torch.py:430-434: default U[0,1) columns (domain is ignored, trap 3 of SURVEY 8a),
        or `sampler.sample(batch_size)`. Products of independent uniform / normal / constant columns -- the default
        sampler and every `NumpySampler(...) & ...` of the reference's examples -- are drawn in HBM by ONE launch of the
        Philox kernel (pinn_sample_points); other samplers through `sample_device` or, last, numpy + a copy. """
        total = self.model.total
        columns = [(engine.SAMPLE_UNIFORM, 0.0, 1.0)] * total if sampler is None else \
            (sampler.columns() if hasattr(sampler, 'columns') else None)
        if columns is not None and len(columns) == total and total <= engine.MAX_INPUTS:
            if self._sample_seed is None:
                self._new_sample_seed()
            seed, call = self._sample_seed, self._sample_calls
            own = sampler.device_key() if hasattr(sampler, 'device_key') else None
            if own is not None:
                # `NumpySampler(..., seed=k)`: the reference draws from THAT seeded generator (model_torch.py:433), whatever
                # torch's seed is -- the sampler's seed keys the Philox stream and the sampler counts its own batches
                rank, _ = self._world()
                seed, call = (own + 7919 * rank) & (2 ** 64 - 1), sampler.next_device_call()
            xs = torch.empty((batch_size, total), dtype=torch.float32, device=self.device)
            self.model.net.sample_points(xs, columns, seed, call, stream=stream)
            self._sample_calls += 1
            return xs
        if self._generator is None:
            rank, _ = self._world()
            self._generator = torch.Generator(device=self.device)
            self._generator.manual_seed(torch.initial_seed() + 7919 * rank)
        if hasattr(sampler, 'sample_device'):
            xs = sampler.sample_device(batch_size, self.device, self._generator)
        else:
            xs = torch.from_numpy(np.asarray(sampler.sample(batch_size)).astype(np.float32)).to(self.device)
        if xs.shape[1] != total:
            raise ValueError(f'sampler produced {xs.shape[1]} columns, the problem has {total}')
        return xs.contiguous()

    # ---- data parallelism ---------------------------------------------------------------------------------
    def begin_data_parallel(self):
        """ once per process group: parameters of rank 0 everywhere, and a communicator whose all-reduce is enqueued on the
        COMPUTE stream (RCCL called directly -- no side stream, no event hops between the kernels of an iteration) """
        _, world = self._world()
        dist = torch.distributed
        if not (dist.is_available() and dist.is_initialized()):
            return
        if not self._broadcast_done:
            dist.broadcast(self.model.flat, src=0)
            self._broadcast_done = True
        if self._comm is None:
            from . import comm
            self._comm = comm.Communicator(self.device)

    def end_data_parallel(self):
        if self._comm is not None:
            self._comm.close()
            self._comm = None

    def _all_reduce(self, stream=None):
        if self._comm is None:
            self.begin_data_parallel()
        if self._comm is not None:              # (no process group: single process, nothing to exchange)
            self._comm.all_reduce_(self.grads, stream)

    def _dp_step(self, xs, world, stream=None, loss_out=None):
        """ one data-parallel iteration on this rank's shard `xs`: tile kernel + reduction, all-reduce of the flat gradient
        buffer (network, log_scale, loss slot, V slots) on the same stream, ONE Adam launch that also records the loss """
        self._fused_step(xs, world, stream=stream)
        self._all_reduce(stream)
        self.optimizer.step(self.grads, loss_out=loss_out, stream=stream)

    def _unreached_scalars(self, offsets, loss_terms, nums_constraints, criterion):
        """ offsets of the trainable scalars (log_scale, V(...) slots) that the loss of these terms does not depend on -- the equation's
        variable in a constraint-only call, log_scale without an initial condition: one dry evaluation of the terms on a few random points
        through the generic step (torch autograd tells), read back once per fit call of a model with variables (FlatAdam.refresh). """
        model = self.model
        if not any(term == 'equation' or 'constraint' in term for term in loss_terms):
            return []
        keep = self.grads.clone()
        try:
            gen = torch.Generator(device=self.device)           # (its own stream: the solver's sampling seeds come from torch's global one)
            gen.manual_seed(5)
            self._generic_step(torch.rand((5, model.total), device=self.device, generator=gen), loss_terms,
                               [num for num in nums_constraints if num < len(self.constraints)], criterion, 1)
            values = self.grads[torch.tensor(offsets, device=self.device)].tolist()
        except Exception:                                       # (an equation that cannot be evaluated on U[0, 1) points: the fit itself will say;
            return []                                           #  every scalar counts as reached, the behaviour before this check existed)
        finally:
            self.grads.copy_(keep)
        return [off for off, v in zip(offsets, values) if v == 0.0]

    def fit(self, niters, batch_size, sampler=None, loss_terms='equation', optimizer='Adam',
            criterion=nn.MSELoss(), lr=0.005, **kwargs):
        """ reference model_torch.py:364-464. Under torch.distributed `batch_size` stays the GLOBAL number of points per
        iteration (the reference's meaning); rank r steps on its share of it. """
        with self._device_guard():
            return self._fit(niters, batch_size, sampler, loss_terms, optimizer, criterion, lr, **kwargs)

    def _fit(self, niters, batch_size, sampler, loss_terms, optimizer, criterion, lr, **kwargs):
        model = self.model
        for num in self._constraints_seen:          # variables an earlier fit call brought to life are trainable now
            model.dormant_variables -= self._born_in_constraint.get(num, set())
        if optimizer is not None:                                                          # :419-422
            plain_adam = optimizer == 'Adam' and not kwargs.get('weight_decay') and not kwargs.get('amsgrad') and \
                not (set(kwargs) - {'betas', 'eps', 'weight_decay', 'amsgrad'})
            self.optimizer = (FlatAdam(model, lr=lr, **kwargs) if plain_adam
                              else TorchOptimizerAdapter(model, optimizer, lr, **kwargs))
        elif self.optimizer is None:
            raise ValueError('optimizer=None reuses the optimizer of a previous fit call; there is none yet')
        self._generic_graph = None          # launch graphs of the generic step are recorded per fit call (closure constants may have changed)
        model.train()
        loss_terms = loss_terms if isinstance(loss_terms, (tuple, list)) else (loss_terms, )
        nums_constraints = [int(term.replace('constraint', '').replace('_', ''))
                            for term in loss_terms if 'constraint' in term]
        self._refresh_traces(set(nums_constraints))
        if isinstance(self.optimizer, FlatAdam) and model.variables:
            # (trainable V(...) scalars: which of them the terms of THIS call reach decides what torch's Adam does with them; one dry
            #  evaluation per fit call. Without variables nothing changes from call to call: log_scale is reached iff there is an IC)
            lay = model.net.layout
            scalars = [lay.off_log_scale] + [off + i for off, n in model.variables.values() for i in range(n)]
            self.optimizer.refresh(self._unreached_scalars(scalars, loss_terms, nums_constraints, criterion))
            if optimizer is None and self.optimizer.lagging(scalars):
                self.optimizer = TorchOptimizerAdapter.continuing(self.optimizer)
        else:
            self.optimizer.refresh()
        self._constraints_seen |= {num for num in nums_constraints if num < len(self.constraints)}
        mse_mean = isinstance(criterion, nn.MSELoss) and criterion.reduction == 'mean'
        lowered = all(num < len(self.constraint_plans) and self.constraint_plans[num] is not None
                      for num in nums_constraints)
        only_known_terms = all(term == 'equation' or 'constraint' in term for term in loss_terms)
        fused = (self.use_fused and mse_mean and not self.ic_trainable and only_known_terms and lowered
                 and len(loss_terms) > 0 and (self.program is not None or 'equation' not in loss_terms))
        rank, world = self._world()
        if world > 1:
            self.begin_data_parallel()
            # every rank must run the same step path (the validation of a lowering is a float comparison on random points and
            # may come out differently on one rank): fused only if fused everywhere
            flag = torch.tensor([1 if fused else 0], dtype=torch.int32, device=self.device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            fused = bool(int(flag.item()))
        local_batch = batch_size // world + (1 if rank < batch_size % world else 0)
        self._global_batch = batch_size
        self._new_sample_seed()
        lay = model.net.layout
        history = torch.zeros(niters, dtype=torch.float32, device=self.device)
        done = [0]
        self._pending.append((history, done))       # registered up front: an interrupted fit keeps the losses it reached
        self.last_fit_path = 'fused' if fused else 'generic'
        flat_adam = isinstance(self.optimizer, FlatAdam)
        one_launch = fused and world == 1 and flat_adam and tuple(loss_terms) == ('equation',)
        stream = engine.stream_of(model.flat)           # looked up once per call, not per iteration
        history_ptr = history.data_ptr()
        try:
            self._fit_loop(niters, local_batch, sampler, loss_terms, nums_constraints, criterion, world, fused, one_launch,
                           flat_adam, stream, history, history_ptr, done)
        finally:
            self._global_batch = None

    def _fit_loop(self, niters, local_batch, sampler, loss_terms, nums_constraints, criterion, world, fused, one_launch,
                  flat_adam, stream, history, history_ptr, done):
        lay = self.model.net.layout
        columns = self._device_columns(sampler) if one_launch else None
        if columns is not None and not self._needs_ic_streams():
            # the common case end to end on the device: chunks of iterations enqueued by ONE library call each
            # (pinn_fit_steps: sample, fused step, Adam per iteration; the interpreter is out of the per-iteration path)
            return self._fit_chunks(niters, local_batch, sampler, columns, stream, history, done)
        for it in tqdm(range(niters), disable=None):
            xs = self._sample(local_batch, sampler, stream)
            if one_launch:
                # Adam rides in the gradient-reduction launch, which also drops the loss into history[it]
                self._fused_step(xs, 1, adam=self.optimizer, loss_out=history_ptr + 4 * it, stream=stream)
                done[0] = it + 1
                continue
            if fused:
                # (two to four launches per iteration: replaying them as a launch graph measured 0.048 -> 0.046 ms/it at batch 500 and
                #  0.072 -> 0.081 at 65 536 -- the eager loop stays)
                self._fused_terms_step(xs, loss_terms, nums_constraints, world, stream=stream)
            else:
                self._generic_step_auto(xs, loss_terms, nums_constraints, criterion, world, remaining=niters - it)
            if world > 1:
                self._all_reduce(stream)                # flat [p_total]: network, log_scale, loss slot, V slots
            if flat_adam:
                self.optimizer.step(self.grads, loss_out=history_ptr + 4 * it, stream=stream)
            else:
                self.optimizer.step(self.grads)
                history[it:it + 1].copy_(self.grads[lay.off_loss:lay.off_loss + 1])
            done[0] = it + 1

    FIT_CHUNK = 128             # iterations per pinn_fit_steps call (progress bar / KeyboardInterrupt granularity)
    FIT_CTRL_ON_HOST = False    # (tests on the CPU emulator: the control block of pinn_fit_steps_graph in host memory)
    GRAPH_MAX_BATCH = 4096      # batches up to this size replay their chunks as one launch graph (+16 % at batch 100, +1 % at 4 096)

    def _device_columns(self, sampler):
        """ (kind, a, b) per input column if the Philox kernel can draw this sampler's batches, else None """
        total = self.model.total
        columns = [(engine.SAMPLE_UNIFORM, 0.0, 1.0)] * total if sampler is None else \
            (sampler.columns() if hasattr(sampler, 'columns') else None)
        if columns is not None and len(columns) == total and total <= engine.MAX_INPUTS:
            return columns
        return None

    def _needs_ic_streams(self):
        model = self.model
        lowered_ic = self.residual_plan is not None and self.residual_plan.ic_row is not None
        return (model.initial_condition is not None and model.ic_constant is None and self.ic_var_slot is None
                and not lowered_ic)

    def _fit_chunks(self, niters, batch, sampler, columns, stream, history, done):
        model, spec, adam = self.model, self.spec, self.optimizer
        comb_w = self.residual_plan.comb_w if self.residual_plan is not None else None
        n2 = spec.n2p if comb_w is None else 1
        ws = model.workspace(batch, spec.nd, spec.n2p)
        # (the batch buffer is kept per shape: its address is part of what a recorded chunk carries)
        xs = self.__dict__.setdefault('_fit_xs', {}).get((batch, model.total))
        if xs is None or xs.device != model.flat.device:
            xs = self._fit_xs[(batch, model.total)] = torch.empty((batch, model.total), dtype=torch.float32, device=self.device)
        own = sampler.device_key() if (sampler is not None and hasattr(sampler, 'device_key')) else None
        rank, _ = self._world()
        # small batches (the latency regime): each chunk as ONE replayable launch graph; the control block the kernels read their
        # per-iteration values from belongs to the solver (include/pinn.h pinn_fit_steps_graph)
        ctrl = None
        if batch <= self.GRAPH_MAX_BATCH and (xs.is_cuda or self.FIT_CTRL_ON_HOST) and os.environ.get('PYDENS_AMD_FIT_GRAPH', '1') != '0':
            if getattr(self, '_fit_ctrl', None) is None or self._fit_ctrl.device != xs.device:
                self._fit_ctrl = torch.zeros(int(model.net.lib.pinn_fit_ctrl_bytes()) + 64, dtype=torch.uint8, device=xs.device)
            ctrl = self._fit_ctrl
        with tqdm(total=niters, disable=None) as bar:
            it = 0
            while it < niters:
                k = min(self.FIT_CHUNK, niters - it)
                if own is not None:         # `NumpySampler(..., seed=k)`: the sampler's own key and batch counter (see _sample)
                    seed = (own + 7919 * rank) & (2 ** 64 - 1)
                    call0 = sampler.next_device_call(0)
                else:
                    seed, call0 = self._sample_seed, self._sample_calls
                t0, applied = adam.t, k
                try:
                    model.net.fit_steps(self.program, model.flat, xs, columns, seed, call0, self.grads, ws, adam.exp_avg,
                                        adam.exp_avg_sq, adam.mask, adam.step_count, adam.t + 1, adam.lr, adam.betas, adam.eps,
                                        history[it:it + k], k, dir_cols=spec.dir_cols, n2=n2, ic_const=model.kernel_ic_const(),
                                        stream=stream, ctrl=ctrl)
                    if os.environ.get('PYDENS_AMD_FIT_PERSIST') == '1' and model.net.lib.pinn_fit_chunk_status() != 0:
                        # the opt-in GRID form of the one-launch chunk gave up at its device-wide wait: nothing of the chunk was applied
                        # (include/pinn.h pinn_fit_chunk_status; the check synchronises with the device once per chunk)
                        raise RuntimeError('libpinn: ' + model.net.lib.pinn_last_error().decode())
                except BaseException:
                    # the library stopped inside the chunk (or an interrupt landed around the call): the device knows how many
                    # Adam steps it applied -- the host's step number and the batch counters follow IT, so that a later
                    # fit(optimizer=None) continues the bias correction and the point stream where the device left off
                    applied = max(0, min(k, int(adam.step_count.item()) - t0))
                    raise
                finally:
                    adam.t = t0 + applied
                    if own is not None:
                        sampler.next_device_call(applied)
                    else:
                        self._sample_calls += applied
                it += k
                done[0] = it
                bar.update(k)

    def _fused_step(self, xs, world, adam=None, loss_out=None, stream=None):
        model, spec = self.model, self.spec
        comb_w = self.residual_plan.comb_w if self.residual_plan is not None else None
        n2 = spec.n2p if comb_w is None else 1              # combined second-order stream: [u, firsts, sum_k c_k u_kk]
        ic_streams = None
        lowered_ic = self.residual_plan is not None and self.residual_plan.ic_row is not None
        if model.initial_condition is not None and model.ic_constant is None and self.ic_var_slot is None and not lowered_ic:
            ic_streams = self._ic_stream_tensor(xs, comb_w)
        ws = model.workspace(xs.shape[0], spec.nd, spec.n2p)
        if adam is not None:
            adam.t += 1
            model.net.residual_adam_step(self.program, model.flat, xs, self.grads, ws, adam.exp_avg, adam.exp_avg_sq,
                                         adam.mask, adam.step_count, adam.t, adam.lr, adam.betas, adam.eps,
                                         dir_cols=spec.dir_cols, n2=n2, ic_streams=ic_streams,
                                         ic_const=model.kernel_ic_const(), loss_out=loss_out, stream=stream)
            return
        # data parallel: this rank's share of a GLOBAL batch (shares may differ by one point: the true global count divides)
        n_global = self._global_batch if (world > 1 and getattr(self, '_global_batch', None)) else xs.shape[0] * world
        model.net.residual_step(self.program, model.flat, xs, self.grads, ws, spec.dir_cols, n2,
                                ic_streams=ic_streams, ic_const=model.kernel_ic_const(),
                                inv_n_global=1.0 / n_global, stream=stream)

    GENERIC_GRAPH_WARMUP = 3    # eager steps in front of the recording (instantiations, workspaces, autograd buffers settle)
    GENERIC_GRAPH_MIN_REPLAYS = 48      # a recording (device synchronise, gc, allocator trim inside torch.cuda.graph) is only made when at
                                        # least this many iterations of the fit call are left to replay it (short fits in a loop)
    GENERIC_GRAPH_CHECK_EVERY = 64      # every this many replays the step is ALSO run eagerly on the same batch and compared
    GENERIC_GRAPH_EARLY_CHECKS = (2, 8, 32)     # ... and soon after the recording, backing off (a recording that froze something is
                                                # noticed within a couple of iterations instead of 64; the steady state pays one check per 64)
    GENERIC_GRAPH_CHECK_RTOL = 1e-5     # relative L2 difference of the gradient buffers that still counts as "the same step" (torch code of
                                        # the user's that sums with atomics -- index_add / scatter backward -- differs in the last bits from run
                                        # to run; the library's own kernels are bit-repeatable)

    def _graph_step(self, xs, key, run, enabled=True, remaining=None):
        """ `run(points)` -- the gradient part of one iteration, everything between drawing the batch and the optimizer step -- replayed
        as ONE launch graph: recorded after a few eager iterations on a static copy of the batch (torch.cuda.graph; the library's
        launches go to the current stream, i.e. into the recording) and replayed with the next batch copied in. Same kernels, same
        order: bit-identical. A step whose torch code cannot be captured (data-dependent shapes, host reads) stays eager for good.

        What a recording FREEZES: the user's equation / constraint Python runs while the graph is recorded and not again -- host-side
        state it reads (closure scalars, numpy RNG used for arithmetic) keeps the value it had then, until the next fit call (graphs
        are per fit call). The reference re-evaluates that Python every iteration. Guard: 2, 8 and 32 replays after the recording and
        every GENERIC_GRAPH_CHECK_EVERY replays from then on the same batch is also stepped eagerly and the gradient buffers are compared
        (equal, or within GENERIC_GRAPH_CHECK_RTOL in relative L2: the library's kernels are bit-repeatable, the user's torch code need
        not be); on a mismatch the eager result is kept and the fit goes on eagerly. Up to GENERIC_GRAPH_CHECK_EVERY - 1 iterations can
        therefore run on frozen host-side state before it is noticed.
        `PYDENS_AMD_STEP_GRAPH=0` turns recording off altogether. """
        st = getattr(self, '_generic_graph', None)
        if not (enabled and xs.is_cuda and os.environ.get('PYDENS_AMD_STEP_GRAPH', os.environ.get('PYDENS_AMD_GENERIC_GRAPH', '1')) != '0'
                and not (st and st.get('failed') and st['key'] == key)):
            return run(xs)
        if st is None or st['key'] != key:
            st = self._generic_graph = dict(key=key, count=0, graph=None, xs=None, failed=False, replays=0)
        if st['graph'] is None:
            if st['count'] < self.GENERIC_GRAPH_WARMUP:
                st['count'] += 1
                before = tokens.AUTOGRAD_FALLBACKS[0]
                out = run(xs)
                if tokens.AUTOGRAD_FALLBACKS[0] != before:
                    # D(...) on something that is not a kernel stream (a constraint that differentiates the model at its point, a
                    # function of the inputs): torch.autograd.grad(create_graph=True) inside the step -- never recorded (see
                    # _generic_step_auto)
                    st['failed'], st['error'] = True, 'the step differentiates inside its torch code (D by autograd): kept eager'
                return out
            if remaining is not None and remaining < self.GENERIC_GRAPH_MIN_REPLAYS:
                return run(xs)                  # (too few iterations left to pay for a recording)
            try:
                st['xs'] = xs.clone()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    run(st['xs'])
                st['graph'] = graph
            except Exception as exc:         # noqa: BLE001 -- whatever the step's torch code does that a capture refuses
                st['failed'], st['graph'], st['error'] = True, None, repr(exc)
                torch.cuda.synchronize()
                return run(xs)
        st['xs'].copy_(xs)
        st['graph'].replay()
        st['replays'] += 1
        if st['replays'] in self.GENERIC_GRAPH_EARLY_CHECKS or st['replays'] % self.GENERIC_GRAPH_CHECK_EVERY == 0:
            replayed = self.grads.clone()
            run(xs)
            if not torch.equal(replayed, self.grads):
                rel = float((replayed - self.grads).norm() / self.grads.norm().clamp_min(1e-30))
                if not rel <= self.GENERIC_GRAPH_CHECK_RTOL:            # (also: NaN)
                    import warnings
                    warnings.warn('pydens_amd: the replayed launch graph and the eager step differ on the same batch (relative L2 '
                                  f'{rel:.1e}): host-side state the equation / constraint code reads may have changed during fit, or its '
                                  'torch code is not reproducible from run to run; continuing without the launch graph', RuntimeWarning)
                    st['failed'], st['graph'], st['error'] = True, None, f'replay differs from the eager step (relative L2 {rel:.1e})'

    def _generic_step_auto(self, xs, loss_terms, nums_constraints, criterion, world, remaining=None):
        """ the generic step -- pinn_jet_forward -> the user's torch code and its autograd sweep (a few dozen small kernels the
        interpreter launches one by one) -> pinn_jet_backward, constraint terms (the model on fixed points) included -- as a launch
        graph in a single process (direction groups included). Kept eager: data-parallel steps (the all-reduce sits between the halves)
        and every step that differentiates INSIDE its torch code with create_graph -- a callable initial condition's derivative
        streams, an equation that takes torch gradients with respect to its inputs, a model subclass's own forward(): recording those
        ends in a crash inside hipStreamEndCapture on ROCm 7.2 (not an exception: tools/graph_crash_probe.py), so they are never tried. """
        model = self.model
        ok = (world == 1 and not self.custom_forward and not self.needs_x_grad
              and not (model.initial_condition is not None and model.ic_constant is None))
        return self._graph_step(xs, ('generic', tuple(xs.shape), id(criterion), id(self._eq), tuple(loss_terms)),
                                lambda pts: self._generic_step(pts, loss_terms, nums_constraints, criterion, world), enabled=ok,
                                remaining=remaining)

    def _fused_terms_step(self, xs, loss_terms, nums_constraints, world, stream=None):
        """ summed loss (:441-457) on the fused kernels: the equation term stores gradient + loss, every constraint term adds its own """
        first = True
        if 'equation' in loss_terms:
            self._fused_step(xs, world, stream=stream)
            first = False
        for num in nums_constraints:
            self._constraint_step(num, world, accumulate=not first)
            first = False

    def _generic_step(self, xs, loss_terms, nums_constraints, criterion, world):
        model, spec = self.model, self.spec
        lay = model.net.layout
        self.grads.zero_()
        model.grad_sink = self.grads
        for name in model.variables:
            getattr(model, name).grad = None
        model.log_scale.grad = None
        try:
            loss = 0
            leaf = None
            # data parallel: the all-reduce SUMS the ranks' buffers, so the equation term (a mean over this rank's share of
            # the global batch) is weighted by that share and the constraint terms, which every rank evaluates, by 1 / world
            n_global = self._global_batch if (world > 1 and getattr(self, '_global_batch', None)) else xs.shape[0] * world
            w_eq, w_con = xs.shape[0] / n_global, 1.0 / world
            kpts = xs                              # where the kernels evaluate the network: the batch, or what a custom forward() maps it to
            if 'equation' in loss_terms:
                if self.custom_forward:
                    self._imap = self._input_map(xs)
                    if self._imap is not None:
                        kpts = self._imap[0]
                if len(spec.groups) == 1:
                    leaf = model.net.jet_forward(model.flat, kpts, spec.dir_cols, spec.n2p,
                                                 ic_const=model.kernel_ic_const()).requires_grad_()
                else:
                    # more directions than one kernel call carries: one forward per group of directions (u comes with each)
                    leaf = torch.empty((spec.n_streams, xs.shape[0]), dtype=torch.float32, device=self.device)
                    for num, (dirs_g, n2g, idx) in enumerate(spec.groups):
                        part = model.net.jet_forward(model.flat, kpts, dirs_g, n2g, ic_const=model.kernel_ic_const())
                        leaf.index_copy_(0, self._group_rows(num, idx), part)
                    leaf.requires_grad_()
                ic_streams = None
                if model.initial_condition is not None and model.ic_constant is None and not self.custom_forward:
                    ic_streams = self._ic_streams(xs, create_graph=True)       # (custom forward(): the IC is torch code of the model)
                r = self._eval_equation(leaf, xs, ic_streams)
                term = criterion(r, torch.zeros_like(xs[:, :1]))                        # :448
                loss = loss + (term if world == 1 else term * w_eq)

            def _forward(*pts):                                                          # :451-454
                return model(self._points_on_device(pts))

            cols = [xs[:, c:c + 1] for c in range(model.total)]
            for num in nums_constraints:                                                  # :456-457
                term = criterion(self.ctx.run(self.constraints[num], _forward, *cols), torch.zeros(1, device=self.device))
                loss = loss + (term if world == 1 else term * w_con)
            loss.backward()
            if leaf is not None and leaf.grad is not None:
                if len(spec.groups) == 1:
                    ws = model.workspace(xs.shape[0], spec.nd, spec.n2p)
                    model.net.jet_backward(model.flat, kpts, leaf.grad.contiguous(), self.grads, ws, spec.dir_cols, spec.n2p,
                                           ic_const=model.kernel_ic_const(), accumulate=True)
                else:
                    # the parameter gradient is linear in the upstream stream gradients: one backward per group, the
                    # gradient of u itself rides with the first group only
                    for num, (dirs_g, n2g, idx) in enumerate(spec.groups):
                        gin = leaf.grad.index_select(0, self._group_rows(num, idx))
                        if num > 0:
                            gin[0].zero_()
                        ws = model.workspace(xs.shape[0], len(dirs_g), n2g)
                        model.net.jet_backward(model.flat, kpts, gin, self.grads, ws, dirs_g, n2g,
                                               ic_const=model.kernel_ic_const(), accumulate=True)
            for name, (off, n) in model.variables.items():
                p = getattr(model, name)
                if p.grad is not None:
                    self.grads[off:off + n] += p.grad.reshape(-1)
                    p.grad = None
            if self.custom_forward and model.log_scale.grad is not None:
                # (the ansatz of a custom forward() is torch code: its log_scale gradient comes from autograd, not from the kernels)
                self.grads[lay.off_log_scale] += model.log_scale.grad.reshape(())
                model.log_scale.grad = None
            self.grads[lay.off_loss] = loss.detach()
        finally:
            model.grad_sink = None
            self._imap = UNSET

    def predict(self, *xs):
        """ reference model_torch.py:466-487 -> ndarray [N,1]. """
        with self._device_guard():
            return self._predict(*xs)

    def _predict(self, *xs):
        model = self.model
        pts = self.reshape_and_concat(xs, device=self.device).contiguous()
        model.eval()
        with torch.no_grad():
            if self.custom_forward:
                u = self.ctx.run(model.forward, pts).view(-1, 1)          # bare network on the kernels, the rest is the model's torch code
            else:
                u = model.net.jet_forward(model.flat, pts, ic_const=model.kernel_ic_const()).view(-1, 1)
                if model.initial_condition is not None and model.ic_constant is None:
                    u = u + self.ctx.run(model.ic_values, pts).expand_as(u)
        return u.detach().cpu().numpy()
