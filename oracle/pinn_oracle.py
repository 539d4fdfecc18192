""" TEST INFRASTRUCTURE ONLY -- CPU restatement ("port") of the reference PINN step.

Restates, in self-contained PyTorch-CPU code, the algorithm of the reference hot path so that it
can travel to the GPU box (where /root/reference does not exist):

    * network builder          <- batchflow `Block` for layouts of 'f'/'a' (reference model_torch.py:164-168)
    * ansatz `anzatc`           <- reference model_torch.py:107-128
    * differentiation token `D` <- reference model_torch.py:174-178 (nested autograd, create_graph=True)
    * trainable token `V`       <- reference model_torch.py:180-188
    * `reshape_and_concat`      <- reference model_torch.py:327-362
    * one `fit` iteration       <- reference model_torch.py:426-464 (MSELoss vs zeros, backward, Adam)
    * `predict`                 <- reference model_torch.py:466-487

The arithmetic is the reference's: the same ATen CPU ops in the same order, fp32 by default. With
`dtype=torch.float64` the very same code is the fp64 arbiter of SURVEY.md section 8c rule 5.

It is pinned against the unmodified reference file by `tests/test_oracle_vs_golden.py`
(fixtures from `oracle/make_golden.py`). Never imported by the product package.
"""
from contextvars import ContextVar, copy_context

import numpy as np
import torch
from torch import nn
from torch.autograd import grad

current_model = ContextVar("oracle_current_model")


def D(y, x):
    """ reference model_torch.py:174-178 """
    return grad(y.sum(), x, retain_graph=True, create_graph=True)[0]


def V(name, *args, **kwargs):
    """ reference model_torch.py:180-188 """
    model = current_model.get()
    if not hasattr(model, name):
        setattr(model, name, nn.Parameter(*args, **kwargs))
    return getattr(model, name)


def _shim_block():
    """ the batchflow stand-in's Block (oracle/batchflow_shim, the same class the reference is run with) """
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'batchflow_shim', 'batchflow', 'models', 'torch',
                        '__init__.py')
    spec = importlib.util.spec_from_file_location('_oracle_batchflow_block', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Block


def build_mlp(n_in, layout, features, activation):
    """ 'f' -> nn.Linear (PyTorch default init), 'a' -> activation module, 'R' / '+' skip connection; spaces ignored. """
    return _shim_block()(inputs=torch.zeros((2, n_in)), layout=layout, features=list(features), activation=activation)


class OracleModel(nn.Module):
    """ TorchModel + ConvBlockModel of the reference (model_torch.py:17-172), fc layouts only. """
    def __init__(self, ndims, initial_condition=None, boundary_condition=None, domain=(0, 1), nparams=0,
                 layout='fafaf', features=(20, 30, 1), activation='Sigmoid', dtype=torch.float32, **kwargs):
        super().__init__()
        features = kwargs.pop('units', features)
        self.dtype = dtype
        self.ndims = ndims
        self.ndims_spatial = ndims if initial_condition is None else ndims - 1      # :25
        self.nparams = nparams
        self.total = ndims + nparams
        if initial_condition is None:
            self.initial_condition = None
        else:                                                                         # :31-35
            self.initial_condition = (initial_condition if callable(initial_condition)
                                      else lambda *args: torch.tensor(initial_condition, dtype=dtype))
        self.boundary_condition = boundary_condition
        if isinstance(domain, (tuple, list)):                                         # :37-46
            if isinstance(domain[0], (float, int)):
                domain = [domain] * ndims
            elif isinstance(domain[0], (tuple, list)):
                pass
            else:
                raise ValueError('Should be either 1d or 2d-sequence of float/ints.')
        else:
            raise ValueError('Should be either 1d or 2d-sequence of float/ints.')
        self.domain = domain
        self.log_scale = nn.Parameter(torch.tensor(0.0, requires_grad=True))         # :50
        self.conv_block = build_mlp(self.total, layout, list(features), activation)
        if dtype != torch.float32:
            self.to(dtype)

    def linears(self):
        return [m for m in self.conv_block if isinstance(m, nn.Linear)]

    def anzatc(self, u, xs):
        """ Hard binding of BC / IC, same ATen op sequence as reference model_torch.py:107-128
        (so fp32 rounding is identical): BC transform first (:118-121), IC transform second (:124-127). """
        nsp = self.ndims_spatial
        sp = xs[:, :nsp]
        t = xs[:, self.ndims - 1:self.ndims]                                # time = column ndims-1 (:111)
        lo = torch.tensor([d[0] for d in self.domain][:nsp], dtype=self.dtype).reshape(1, -1)
        hi = torch.tensor([d[1] for d in self.domain][:nsp], dtype=self.dtype).reshape(1, -1)
        t0 = self.domain[-1][0]                                             # (:115)
        if self.boundary_condition is not None:
            rise = torch.prod((sp - lo) / (hi - lo), dim=1, keepdim=True)
            fall = torch.prod((hi - sp) / (hi - lo), dim=1, keepdim=True)
            u = u * (rise * fall) + self.boundary_condition
        if self.initial_condition is not None:
            gate = torch.sigmoid((t - t0) / torch.exp(self.log_scale)) - .5
            cols = [sp[:, i] for i in range(nsp)]                           # IC sees 1-D [N] columns (:125)
            u = gate * u + self.initial_condition(*cols).view(-1, 1)
        return u

    def forward(self, xs):
        return self.anzatc(self.conv_block(xs), xs)


class OracleSolver:
    """ reference Solver (model_torch.py:191-487) restated; `dtype` selects fp32 (reference) or fp64 (arbiter). """
    def __init__(self, equation, ndims, initial_condition=None, boundary_condition=None, domain=(0, 1),
                 nparams=0, constraints=None, dtype=torch.float32, model=None, **kwargs):
        self.equation = equation
        self.dtype = dtype
        if constraints is None:
            self.constraints = ()
        elif isinstance(constraints, (tuple, list)):
            self.constraints = constraints
        else:
            self.constraints = (constraints, )
        self.losses = []
        self.optimizer = None
        # (`model`: the reference's plug-in seam, model_torch.py:299-313 -- a subclass of the model with its own forward())
        self.model = (model or OracleModel)(**kwargs, ndims=ndims, initial_condition=initial_condition,
                                 boundary_condition=boundary_condition, domain=domain, nparams=nparams, dtype=dtype)
        current_model.set(self.model)
        self.ctx = copy_context()
        xs = [torch.rand((1, 1), dtype=dtype) for _ in range(self.model.total)]      # fake run :320-325
        for x in xs:
            x.requires_grad_()
        u_hat = self.ctx.run(self.model, self.reshape_and_concat(xs))
        _ = self.ctx.run(self.equation, u_hat, *xs)

    def reshape_and_concat(self, tensors):
        """ Input casting rules of reference model_torch.py:327-362: batch = longest array/tensor/list;
        numbers are tiled; an ndarray of another size is replaced by its first element, tiled (:355-356). """
        items = list(tensors)
        lengths = [int(np.prod(np.shape(t))) for t in items if isinstance(t, (np.ndarray, torch.Tensor, tuple, list))]
        n = max(lengths) if lengths else 1
        cols = []
        for x in items:
            if isinstance(x, torch.Tensor):
                col = x.view(-1, 1)
            elif isinstance(x, np.ndarray):
                arr = x if x.size == n else np.tile(x.squeeze()[0], (n, 1))
                col = torch.tensor(arr.reshape(n, 1), dtype=self.dtype)
            elif isinstance(x, (list, tuple)):
                col = torch.tensor(x, dtype=self.dtype).view(-1, 1)
            else:
                col = torch.tensor(np.tile(x, (n, 1)), dtype=self.dtype)
            cols.append(col)
        return torch.cat(cols, dim=1)

    # ---- parameter import/export in the common (W [out,in], b [out]) per layer + log_scale form ------------------
    def export_params(self):
        out = []
        for lin in self.model.linears():
            out += [lin.weight.detach().cpu().numpy().copy(), lin.bias.detach().cpu().numpy().copy()]
        out.append(self.model.log_scale.detach().cpu().numpy().copy())
        return out

    def import_params(self, arrays):
        with torch.no_grad():
            it = iter(arrays)
            for lin in self.model.linears():
                lin.weight.copy_(torch.as_tensor(next(it), dtype=self.dtype))
                lin.bias.copy_(torch.as_tensor(next(it), dtype=self.dtype))
            self.model.log_scale.copy_(torch.as_tensor(next(it), dtype=self.dtype))

    def export_grads(self):
        out = []
        params = [p for lin in self.model.linears() for p in (lin.weight, lin.bias)] + [self.model.log_scale]
        for p in params:
            out.append(None if p.grad is None else p.grad.detach().cpu().numpy().copy())
        return out

    # ---- one evaluation on given points: u_hat, residual, loss, param grads (reference :435-460, no optimizer) ---
    def evaluate(self, points, chunk=None):
        """ points: ndarray [N,total]. Returns dict(u, r, loss) and leaves .grad on the params.
        `chunk` accumulates sum r^2 and grads over slices (autograd memory at large N). """
        points = np.asarray(points)
        n = points.shape[0]
        for p in self.model.parameters():
            p.grad = None
        us, rs, sumsq = [], [], 0.0
        step = chunk or n
        for lo in range(0, n, step):
            pts = torch.tensor(points[lo:lo + step], dtype=self.dtype)
            xs = [pts[:, i:i + 1].clone().requires_grad_() for i in range(pts.shape[1])]
            xs_concat = self.reshape_and_concat(xs)
            u_hat = self.ctx.run(self.model, xs_concat)
            r = self.ctx.run(self.equation, u_hat, *xs)
            part = (r ** 2).sum() / n
            part.backward()
            sumsq += float(part.detach())
            us.append(u_hat.detach().cpu().numpy()); rs.append(r.detach().cpu().numpy())
        return dict(u=np.concatenate(us), r=np.concatenate(rs), loss=sumsq)

    def fit(self, niters, batch_size, sampler=None, loss_terms='equation', optimizer='Adam',
            criterion=nn.MSELoss(), lr=0.005, points=None, **kwargs):
        """ reference model_torch.py:364-464. `points` (ndarray [niters, N, total]) replaces the sampler
        with a fixed stream of batches so that two engines can be stepped on identical data. """
        if optimizer is not None:
            self.optimizer = getattr(torch.optim, optimizer)([p for p in self.model.parameters() if p.requires_grad],
                                                             lr=lr, **kwargs)
        self.model.train()
        for it in range(niters):
            self.optimizer.zero_grad()
            if points is not None:
                arr = np.asarray(points[it])
                xs = [torch.tensor(arr[:, i:i + 1], dtype=self.dtype) for i in range(arr.shape[1])]
            elif sampler is None:
                xs = [torch.rand((batch_size, 1), dtype=self.dtype) for _ in range(self.model.total)]
            else:
                xs_array = sampler.sample(batch_size).astype(np.float32)
                xs = [torch.from_numpy(xs_array[:, i:i + 1]).to(self.dtype) for i in range(xs_array.shape[1])]
            for x in xs:
                x.requires_grad_()
            xs_concat = self.reshape_and_concat(xs)
            u_hat = self.ctx.run(self.model, xs_concat)
            loss_terms = loss_terms if isinstance(loss_terms, (tuple, list)) else (loss_terms, )
            nums_constraints = [int(term.replace('constraint', '').replace('_', ''))
                                for term in loss_terms if 'constraint' in term]
            loss = 0
            if 'equation' in loss_terms:
                loss += criterion(self.ctx.run(self.equation, u_hat, *xs), torch.zeros_like(xs[0]))

            def _forward(*xs):
                return self.model(self.reshape_and_concat(xs))

            for num in nums_constraints:
                loss += criterion(self.ctx.run(self.constraints[num], _forward, *xs),
                                  torch.zeros(1, dtype=self.dtype))
            loss.backward()
            self.optimizer.step()
            self.losses.append(loss.detach().cpu().numpy())

    def predict(self, *xs):
        """ reference model_torch.py:466-487 """
        xs = self.reshape_and_concat(xs)
        self.model.eval()
        return self.ctx.run(self.model, xs).detach().cpu().numpy()
