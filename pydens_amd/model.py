""" Problem/model description on the host: same classes, constructor arguments and attributes as the reference's
`TorchModel` / `ConvBlockModel` (pydens/model_torch.py:17-172), but the network lives in ONE flat fp32 device
buffer laid out for the HIP kernels (include/pinn.h); `nn.Parameter`s are views into it, so `parameters()`,
`state_dict()`, `freeze_trainable` and user code that pokes weights keep working. """
from abc import ABC, abstractmethod

import torch
from torch import nn

from . import engine, trace


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError('pydens_amd needs a HIP device (torch.cuda.is_available() is False); '
                           'there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


class TorchModel(ABC, nn.Module):
    """ reference model_torch.py:17-128. """
    def __init__(self, ndims, initial_condition=None, boundary_condition=None, domain=(0, 1), nparams=0, **kwargs):
        _ = kwargs
        super().__init__()
        self.ndims = ndims
        self.ndims_spatial = ndims if initial_condition is None else ndims - 1          # :25
        self.nparams = nparams
        self.total = ndims + nparams
        self.variables = {}
        # variables the reference would not have created yet: a V(...) that appears only inside a constraint comes to life
        # in the first iteration that evaluates the constraint (model_torch.py:457), AFTER that fit call has built its
        # optimizer (:420), so that call does not train it. The solver keeps such names here until then.
        self.dormant_variables = set()
        if initial_condition is None:
            self.initial_condition = None
            self.ic_constant = None
        elif callable(initial_condition):
            self.initial_condition = initial_condition
            self.ic_constant = None
        else:                                                                             # :31-35
            self.ic_constant = float(initial_condition)
            self.initial_condition = lambda *args: torch.tensor(self.ic_constant, dtype=torch.float32)
        self.boundary_condition = boundary_condition
        if isinstance(domain, (tuple, list)):                                             # :37-46
            if isinstance(domain[0], (float, int)):
                domain = [domain] * ndims
            elif isinstance(domain[0], (tuple, list)):
                pass
            else:
                raise ValueError('Should be either 1d or 2d-sequence of float/ints.')
        else:
            raise ValueError('Should be either 1d or 2d-sequence of float/ints.')
        self.domain = domain

    @abstractmethod
    def forward(self, xs):
        """ Forward of the model-network. """

    def anzatc(self, u, xs):
        """ hard binding of the boundary / initial condition in plain torch (reference model_torch.py:107-128: boundary transform
        first, :118-121, initial-condition gate second, :124-127; time is column ndims-1, :111). ConvBlockModel's own forward
        has this inside the HIP kernels; a subclass with its OWN forward() -- `self.anzatc(self.conv_block(xs), xs) * ...` --
        calls it here on the network value the kernels hand over, and `D` differentiates through it by the chain rule over the
        kernels' derivative streams. """
        nsp = self.ndims_spatial
        sp = xs[:, :nsp]
        t = xs[:, self.ndims - 1:self.ndims]
        lo = torch.tensor([d[0] for d in self.domain][:nsp], dtype=torch.float32, device=xs.device).reshape(1, -1)
        hi = torch.tensor([d[1] for d in self.domain][:nsp], dtype=torch.float32, device=xs.device).reshape(1, -1)
        t0 = self.domain[-1][0]
        if self.boundary_condition is not None:
            rise = torch.prod((sp - lo) / (hi - lo), dim=1, keepdim=True)
            fall = torch.prod((hi - sp) / (hi - lo), dim=1, keepdim=True)
            u = u * (rise * fall) + self.boundary_condition
        if self.initial_condition is not None:
            gate = torch.sigmoid((t - t0) / torch.exp(self.log_scale)) - .5
            cols = [sp[:, i] for i in range(nsp)]                           # the IC sees 1-D [N] columns (:125)
            ic = self.initial_condition(*cols)
            ic = ic if isinstance(ic, torch.Tensor) else torch.tensor(float(ic), dtype=torch.float32)
            u = gate * u + ic.to(device=xs.device, dtype=torch.float32).view(-1, 1)
        return u

    def freeze_trainable(self, layers=None, variables=None):
        """ reference model_torch.py:56-80: flips requires_grad; `Solver.fit` turns it into the Adam mask. """
        for layer in (layers or []):
            for param in getattr(self, layer).parameters():
                param.requires_grad = False
        for variable in (variables or []):
            getattr(self, variable).requires_grad = False

    def unfreeze_trainable(self, layers=None, variables=None):
        """ reference model_torch.py:82-105 """
        for layer in (layers or []):
            for param in getattr(self, layer).parameters():
                param.requires_grad = True
        for variable in (variables or []):
            getattr(self, variable).requires_grad = True


class FlatLinear(nn.Module):
    """ fc layer whose weight [out,in] / bias [out] are views into the model's flat kernel buffer. """
    def __init__(self, weight, bias):
        super().__init__()
        self.weight = nn.Parameter(weight)
        self.bias = nn.Parameter(bias)
        self.in_features, self.out_features = weight.shape[1], weight.shape[0]

    def forward(self, x):
        return nn.functional.linear(x, self.weight, self.bias)


def _activation_name(act):
    """ what `getattr(nn, activation)()` / a module class / a module instance / a torch function names (model_torch.py:159, :164-168) ->
    the kernels' activation name (`Name:value` for a LeakyReLU / ELU / Softplus instance with a non-default negative_slope / alpha / beta). """
    if isinstance(act, str):
        return act
    if isinstance(act, type):
        return act.__name__
    if isinstance(act, nn.Module):
        # instances configured away from torch's defaults (round 6): the ONE parameter of LeakyReLU / ELU / Softplus travels with the name
        # as 'Name:value' (engine.Net -> pinn_set_act_params); Softplus' threshold stays 20
        if isinstance(act, nn.Softplus) and float(act.threshold) != 20.0:
            raise NotImplementedError(f'nn.Softplus(threshold={act.threshold}): the HIP kernels implement threshold 20 (any beta)')
        for cls, key, default in ((nn.Softplus, 'beta', 1.0), (nn.LeakyReLU, 'negative_slope', 0.01), (nn.ELU, 'alpha', 1.0)):
            if isinstance(act, cls):
                value = float(getattr(act, key))
                if cls is nn.Softplus and not value > 0.0:
                    raise NotImplementedError(f'nn.Softplus(beta={value}): beta must be positive')
                return cls.__name__ if value == default else f'{cls.__name__}:{value!r}'
        if isinstance(act, nn.GELU) and getattr(act, 'approximate', 'none') != 'none':
            if act.approximate != 'tanh':
                raise NotImplementedError(f'nn.GELU(approximate={act.approximate!r}) is not implemented by the HIP kernels')
            return 'GELU_tanh'
        return type(act).__name__
    functions = {torch.sin: 'Sin', torch.tanh: 'Tanh', torch.sigmoid: 'Sigmoid', torch.nn.functional.softplus: 'Softplus',
                 torch.nn.functional.silu: 'SiLU', torch.nn.functional.gelu: 'GELU', torch.relu: 'ReLU', torch.nn.functional.relu: 'ReLU',
                 torch.nn.functional.leaky_relu: 'LeakyReLU', torch.nn.functional.elu: 'ELU', torch.nn.functional.softsign: 'Softsign',
                 torch.nn.functional.mish: 'Mish', torch.nn.functional.selu: 'SELU', torch.nn.functional.tanhshrink: 'Tanhshrink',
                 torch.nn.functional.logsigmoid: 'LogSigmoid', torch.nn.functional.tanh: 'Tanh', torch.nn.functional.sigmoid: 'Sigmoid'}
    for fn, name in functions.items():
        if act is fn:
            return name
    raise NotImplementedError(f'activation {act!r}: the HIP kernels implement Tanh, Sigmoid, Sin, Softplus, SiLU, GELU (erf and tanh '
                              'forms), ReLU, LeakyReLU, ELU, SELU, Softsign, Mish, Tanhshrink and LogSigmoid (torch defaults)')


def parse_fc_layout(layout, features, activation):
    """ layout string -> (widths, one activation name per hidden layer, skip connections) for the kernels.

    Letters (reference model_torch.py:142-156): 'f' dense layer, 'a' activation (a shared one or a sequence with one
    entry per 'a'), 'R' start / '+' end of a skip connection; spaces are ignored. The net must end in a 1-unit dense
    layer. A hidden 'f' without an 'a' gets the identity. 'R' and '+' sit behind a dense layer, either behind its activation
    ('faR fa fa+': sum of activation outputs) or between the layer and its activation ('faR fa f+a': the usual residual block
    act(W h + skip); 'fRa fa f+a': the pre-activation block, z carried to z). Skips join layers of equal width; they may NEST
    ('fa R fa R fa fa + fa + f': '+' closes the most recent open 'R', the way the letters pair in the reference's Block) as long as at
    most one skip starts and at most one ends at a layer; conv letters are out of scope. Skips are returned as (src, dst, pre, src_pre) hidden-layer indices: the output of
    layer dst -- its pre-activation if `pre` -- gets the output (pre-activation if `src_pre`) of layer src added. """
    letters = layout.replace(' ', '')
    features = list(features)
    if set(letters) - set('faR+'):
        raise NotImplementedError(f"layout {layout!r}: only 'f' (dense), 'a' (activation), 'R' and '+' (skip) are supported")
    n_f = letters.count('f')
    if n_f < 2 or not letters.endswith('f'):
        raise NotImplementedError(f"layout {layout!r}: expected at least two dense layers, the last one (1 unit) at the end")
    if len(features) < n_f:
        raise ValueError(f'layout {layout!r} needs {n_f} entries in `features`, got {features}')
    features = features[:n_f]
    if features[-1] != 1:
        raise NotImplementedError('the last dense layer must have one unit (the scalar solution approximation)')
    act_list = list(activation) if isinstance(activation, (list, tuple)) else None
    if act_list is not None and len(act_list) < letters.count('a'):
        raise ValueError(f'layout {layout!r} needs {letters.count("a")} activations, got {len(act_list)}')
    acts, skips, open_stack, layer = [], [], [], -1
    for letter in letters:
        if letter == 'f':
            if layer >= 0 and len(acts) == layer:
                acts.append('Identity')
            layer += 1
        elif letter == 'a':
            if layer < 0 or len(acts) > layer:
                raise NotImplementedError(f"layout {layout!r}: every 'a' must follow its own dense layer")
            acts.append(_activation_name(act_list.pop(0) if act_list is not None else activation))
        else:
            pre = layer >= 0 and len(acts) == layer            # between a dense layer and its activation: the pre-activation
            if layer < 0 or (len(acts) != layer + 1 and not pre):
                raise NotImplementedError(f"layout {layout!r}: 'R' / '+' must come after a dense layer (in front of or behind its activation)")
            if letter == 'R':
                if any(src == layer for src, _ in open_stack) or any(s[0] == layer for s in skips):
                    raise NotImplementedError(f"layout {layout!r}: two skip connections start at dense layer {layer + 1}")
                open_stack.append((layer, pre))
            else:
                if not open_stack or open_stack[-1][0] == layer:
                    raise NotImplementedError(f"layout {layout!r}: '+' needs an open 'R' with a layer in between")
                open_skip, open_pre = open_stack.pop()             # (brackets: '+' closes the most recent 'R')
                if any(s[1] == layer for s in skips):
                    raise NotImplementedError(f"layout {layout!r}: two skip connections end at dense layer {layer + 1}")
                if features[open_skip] != features[layer]:
                    raise ValueError(f"layout {layout!r}: skip connection joins widths {features[open_skip]} and {features[layer]}")
                skips.append((open_skip, layer, pre, open_pre))
    if open_stack:
        raise ValueError(f"layout {layout!r}: 'R' without a closing '+'")
    return features, acts, skips


class _ModelForward(torch.autograd.Function):
    """ value-only network+ansatz on arbitrary points with gradients to the parameters (constraint terms,
    reference model_torch.py:451-457). Forward = pinn_jet_forward(nd=0); backward = pinn_jet_backward(nd=0),
    accumulated straight into the solver's flat gradient buffer (`model.grad_sink`): the nn.Parameters are views
    of the flat kernel buffer, so the parameter gradient is delivered there and not through autograd leaves. """
    @staticmethod
    def forward(ctx, anchor, xs, model):
        ctx.model, ctx.xs = model, xs
        return model.net.jet_forward(model.flat, xs, ic_const=model.kernel_ic_const()).view(-1, 1)

    @staticmethod
    def backward(ctx, grad_out):
        model, xs = ctx.model, ctx.xs
        if model.grad_sink is None:
            raise RuntimeError('gradients through model(xs) are only available inside Solver.fit')
        ws = model.workspace(xs.shape[0], 0, 0)
        model.net.jet_backward(model.flat, xs, grad_out.contiguous().view(1, -1), model.grad_sink, ws,
                               ic_const=model.kernel_ic_const(), accumulate=True)
        return None, None, None


class InputProbe(Exception):
    """ raised by `model.conv_block` while the solver asks a custom forward() what it hands to the network (Solver._input_map) """
    def __init__(self, arg):
        super().__init__('conv_block input probe')
        self.arg = arg


PROBE = object()        # `model.raw_field` while that question is being asked


class KernelBlock(nn.Sequential):
    """ `model.conv_block` (reference model_torch.py:164-172): holds the layers' parameters as views of the flat kernel buffer; CALLED
    -- which only a subclass with its own forward() does -- it returns the network value computed by the HIP kernels: inside an
    equation evaluation the value stream the solver already has (tagged, so that `D` finds the derivative streams), elsewhere
    (predict, constraints) a value-only kernel forward that is differentiable with respect to the parameters. Inside an equation
    evaluation the argument must be the batch of points itself or a fixed map of it, point by point, onto as many columns (`2 * xs - 1`,
    `torch.sin(xs)`, a time warp, a rotation: the solver asked beforehand -- Solver._input_map --, evaluated the kernels THERE and applies
    the chain rule: scaled streams for per-column affine maps, the map's Jacobian by autograd otherwise); maps that depend on trainable
    parameters, on the batch as a whole or that change the number of columns are refused. """
    def __init__(self, model):
        super().__init__()
        object.__setattr__(self, '_model', model)

    def forward(self, xs):
        model = self._model
        model.conv_block_calls += 1
        field = model.raw_field
        if field is PROBE:
            raise InputProbe(xs)
        if field is not None:
            pts, value, expect = field
            want = pts if expect is None else expect
            if xs is not want:
                # a forward() that hands over a view or a copy of its argument. The SAME MEMORY seen through the same shape and strides
                # (`xs[:, :]`, `xs.view_as(xs)`) is the batch by construction: no compare. A real copy (`xs.clone()`, `torch.cat(cols, 1)`,
                # `2 * xs - 1`) is another buffer whose content only a compare can vouch for -- in every call: the caching allocator hands
                # a fresh temporary the address of the last one, so a remembered verdict would keep accepting `xs * self.scale` after `scale`
                # moved (the compare is an N x d pass + a device sync, paid only by forwards that copy)
                same_memory = (expect is None and xs.data_ptr() == pts.data_ptr() and xs.shape == pts.shape and xs.stride() == pts.stride()
                               and xs.dtype == pts.dtype and xs.device == pts.device)
                if not same_memory and (xs.shape != want.shape or not torch.equal(xs.detach().to(want.dtype), want.detach())):
                    raise NotImplementedError('a custom forward() may call self.conv_block only on the batch of points it was given or on a '
                                              'fixed pointwise map of it (the argument differs from what the same forward() handed over when '
                                              'the solver asked)')
            sc = trace.active_streams.get()
            if sc is not None and sc.ymap is not None:
                sc.ymap['ys'] = xs              # (connected to the solver's columns by the forward()'s own torch ops: the Jacobian comes from there)
            return value
        xs = xs.to(device=model.flat.device, dtype=torch.float32).contiguous()
        return _ModelForward.apply(model._anchor, xs, model)


class ConvBlockModel(TorchModel):
    """ reference model_torch.py:130-172 for fully connected layouts. """
    def __init__(self, ndims, initial_condition=None, boundary_condition=None, domain=(0, 1), nparams=0,
                 layout='fafaf', features=(20, 30, 1), activation='Sigmoid', device=None, _lib=None, **kwargs):
        # (`_lib`: TEST HOOK, not part of the reference's surface -- the CPU test tier hands over the emulator build of the same C-ABI,
        #  tools/ an experiment build; the product loads pydens_amd/libpinn_hip.so itself and refuses anything but the gfx950 backend)
        features = kwargs.pop('units', features)                  # README.md:41-42 spells it `units`
        super().__init__(ndims=ndims, initial_condition=initial_condition, boundary_condition=boundary_condition,
                         domain=domain, nparams=nparams, **kwargs)
        widths, act_names, skips = parse_fc_layout(layout, features, activation)
        self.activation_names, self.skips = act_names, skips
        self.device = torch.device(device) if device is not None else default_device()
        self.layer_dims = [self.total] + widths
        # a subclass with its own forward() (the reference's plug-in seam, model_torch.py:52-54, :312-313): the kernels compute the
        # bare network, the ansatz -- if the subclass calls self.anzatc -- is torch code of its forward()
        self.custom_forward = type(self).forward is not ConvBlockModel.forward
        self.raw_field = None               # (points, tagged network value) while the solver evaluates the equation
        self.conv_block_calls = 0
        bare = self.custom_forward
        self.net = engine.Net(self.layer_dims, act_names, ndims, nparams,
                              has_bc=boundary_condition is not None and not bare, bc_value=(boundary_condition or 0.0) if not bare else 0.0,
                              has_ic=initial_condition is not None and not bare, domain=self.domain, lib=_lib, skips=skips)
        lay = self.net.layout
        # flat kernel buffer; PyTorch-default nn.Linear init drawn in the reference's order
        # (fake inputs first, model_torch.py:167, then the layers of Block, :168)
        host = torch.zeros(lay.p_total, dtype=torch.float32)
        _ = torch.rand((2, self.total), dtype=torch.float32)
        for (w, b), (n_in, n_out) in zip(self.net.param_views(host), zip(self.layer_dims[:-1], self.layer_dims[1:])):
            lin = nn.Linear(n_in, n_out, bias=True)
            w.copy_(lin.weight.detach()); b.copy_(lin.bias.detach())
        self.register_buffer('flat', host.to(self.device), persistent=False)
        self.log_scale = nn.Parameter(self.flat.as_strided((), (), lay.off_log_scale))   # :50, value 0.0
        self.conv_block = KernelBlock(self)
        for i, (w, b) in enumerate(self.net.param_views(self.flat)):
            self.conv_block.add_module(f'fc{i + 1}', FlatLinear(w, b))
        self._next_extra = lay.off_extra
        self._workspaces = {}
        self._anchor = torch.zeros((), device=self.device, requires_grad=True)
        self.grad_sink = None

    # ---- trainable V(...) variables live in the tail of the flat buffer ---------------------------------------
    def register_variable(self, name, param):
        lay = self.net.layout
        n = param.numel()
        if self._next_extra + n > lay.p_total:
            raise RuntimeError(f'V({name!r}): no room left for {n} more trainable scalars '
                               f'({engine.EXTRA_SLOTS} slots per model)')
        off = self._next_extra
        self._next_extra += n
        with torch.no_grad():
            self.flat[off:off + n] = param.detach().to(self.flat).reshape(-1)
        view = self.flat.as_strided(tuple(param.shape), tuple(param.stride()) if param.dim() else (), off)
        self.variables[name] = (off, n)
        return nn.Parameter(view, requires_grad=param.requires_grad)

    # ---- helpers for the solver --------------------------------------------------------------------------------
    def kernel_ic_const(self):
        """ constant initial condition handled inside the kernels (0 when the IC is a callable added by the host) """
        return self.ic_constant if (self.ic_constant is not None and not self.custom_forward) else 0.0

    def workspace(self, n_points, nd, n2):
        key = (nd, n2)
        need = self.net.workspace_bytes(n_points, nd, n2)
        ws = self._workspaces.get(key)
        if ws is None or ws.numel() * 4 < need:
            ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.flat.device)
            self._workspaces[key] = ws
        return ws

    def trainable_mask(self):
        """ uint8 [p_total]: 1 for entries Adam may touch = real (non-padding) parameters with requires_grad. """
        mask = torch.zeros_like(self.flat, dtype=torch.uint8)
        lay = self.net.layout
        lins = [m for m in self.conv_block if isinstance(m, FlatLinear)]
        for (w, b), lin in zip(self.net.param_views(mask), lins):
            w.fill_(1 if lin.weight.requires_grad else 0)
            b.fill_(1 if lin.bias.requires_grad else 0)
        mask[lay.off_log_scale] = 1 if self.log_scale.requires_grad else 0
        for name, (off, n) in self.variables.items():
            mask[off:off + n] = 1 if getattr(self, name).requires_grad and name not in self.dormant_variables else 0
        return mask

    def optimizer_parameters(self):
        """ what the reference hands to the optimizer at the start of a fit call (model_torch.py:420): every parameter
        with requires_grad that EXISTS at that moment (see `dormant_variables`). """
        dormant = {id(getattr(self, name)) for name in self.dormant_variables}
        return [p for p in self.parameters() if p.requires_grad and id(p) not in dormant]

    def ic_values(self, xs):
        """ IC(x_spatial) as [N,1] on the device (callable ICs only; differentiable w.r.t. V-variables). """
        cols = [xs[:, i] for i in range(self.ndims_spatial)]              # 1-D [N] columns, model_torch.py:125
        val = self.initial_condition(*cols)
        if not isinstance(val, torch.Tensor):
            val = torch.tensor(float(val), dtype=torch.float32)
        return val.to(device=xs.device, dtype=torch.float32).view(-1, 1)

    def forward(self, xs):
        """ u_hat [N,1] = anzatc(conv_block(xs), xs) (reference model_torch.py:170-172) on the HIP kernels. """
        xs = xs.to(device=self.flat.device, dtype=torch.float32).contiguous()
        u = _ModelForward.apply(self._anchor, xs, self)
        if self.initial_condition is not None and self.ic_constant is None:
            u = u + self.ic_values(xs).expand_as(u)
        return u
