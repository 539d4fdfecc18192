""" The two tokens user equations are written with: `D` (differentiate) and `V` (trainable variable).
Same names, call signatures and meaning as the reference (pydens/model_torch.py:174-188). """
from contextvars import ContextVar

import torch
from torch import nn

from . import trace

current_model = ContextVar("current_model")          # reference model_torch.py:15


# how often D had to differentiate by torch autograd (create_graph) instead of picking a kernel stream: the solver never records a step
# as a launch graph in which that happens (Solver._graph_step)
AUTOGRAD_FALLBACKS = [0]


def D(y, x):
    """ Differentiation token: per-sample dy/dx with y, x of shape [N,1] (reference model_torch.py:174-178).

    The reference runs a nested reverse sweep through the network for every call. Here the derivatives of the
    solution approximation are *streams* computed by the HIP kernels in the forward pass, so
      * `D(f, x)` / `D(D(f, x), x)` on the field (or on a stream) just picks the stream;
      * `D(expr, x)` of an expression of streams and inputs applies the chain rule
        d expr/dx = (d expr/dx)|explicit + sum_a (d expr/d stream_a) * stream_{a + e_x}, the partials by torch
        autograd over the few pointwise ops of `expr` (never through the network);
      * anything unrelated to the field falls back to the reference's plain autograd formula.
    """
    if isinstance(y, trace.Sym) or isinstance(x, trace.Sym):
        return trace.sym_D(y, x)
    alpha = getattr(y, '_pinn_alpha', None)
    col = getattr(x, '_pinn_col', None)
    sc = trace.active_streams.get()
    if alpha is not None and col is not None:
        return y._pinn_ctx.derivative(alpha, col)
    if sc is not None and col is not None and y.requires_grad:
        sc.used_autograd_fallback = True
        AUTOGRAD_FALLBACKS[0] += 1
        items = [(a, t) for a, t in sc.stream_tensors() if t.requires_grad]
        wrt = [x] + [t for _, t in items]
        grads = torch.autograd.grad(y.sum(), wrt, retain_graph=True, create_graph=True, allow_unused=True)
        total = grads[0] if grads[0] is not None else torch.zeros_like(x)
        for (a, _), g in zip(items, grads[1:]):
            if g is not None:
                total = total + g * sc.derivative(a, col)
        return total
    if sc is not None:
        sc.used_autograd_fallback = True
    AUTOGRAD_FALLBACKS[0] += 1
    return torch.autograd.grad(y.sum(), x, retain_graph=True, create_graph=True)[0]


def V(name, *args, **kwargs):
    """ Token for a trainable variable (reference model_torch.py:180-188): created on first use and registered on
    the model of the current context, fetched afterwards. The storage lives in the model's flat parameter buffer
    so that the HIP Adam kernel updates it together with the network. """
    model = current_model.get()
    if not hasattr(model, name):
        param = nn.Parameter(*args, **kwargs)
        register = getattr(model, 'register_variable', None)
        setattr(model, name, register(name, param) if register is not None else param)
    param = getattr(model, name)
    lookup = trace._variable_slot.get()
    if lookup is not None:
        # symbolic trace of the equation (trace.symbolic): a scalar variable is a register of the residual program, so
        # that expressions of variables alone (`0.1 * V('a') * V('b')`) stay symbolic as well
        slot = lookup(param)
        if slot is not None and slot < trace.MAX_VARS:
            return trace.Sym('var', col=slot)
    return param
