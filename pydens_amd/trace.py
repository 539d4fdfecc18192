""" Equation tracing: from a pydens equation callable to (a) the derivative streams it needs and (b), when the
equation is a plain pointwise expression, a residual program the fused HIP kernel interprets per point.

The reference evaluates `equation(u_hat, *xs)` with torch tensors and lets `D` call nested autograd
(pydens/model_torch.py:174-178, :447-448). Here the kernels deliver u and its derivatives as *streams*
(include/pinn.h), so the host has to know which streams an equation asks for:

  * `discover`  -- runs the callable once on tagged dummy tensors (the same "fake run" the reference does at
                   model_torch.py:319-325) and records every multi-index `D` requests;
  * `symbolic`  -- runs it once more on `Sym` objects that record arithmetic and `torch.*` calls into a DAG,
                   which `compile_program` lowers to the register code of `pinn_program_t`.

If the symbolic run meets anything it cannot express (trainable `V`, tensor constants, unsupported ops), the
caller keeps the generic path: streams from the kernel, the user's own torch code for the residual, upstream
stream gradients back into the kernel.
"""
import math
import numbers
from contextvars import ContextVar

import numpy as np
import torch

from .engine import OPS, MAX_OPS, MAX_CONSTS, MAX_REGS, MAX_DIRS

active_streams = ContextVar('pinn_active_streams', default=None)


class TraceUnsupported(Exception):
    """ the equation cannot be lowered to a residual program (generic path is used instead). """


# ---------------------------------------------------------------------------------------------------------------
# stream bookkeeping
# ---------------------------------------------------------------------------------------------------------------
class StreamSpec:
    """ Which derivative streams the kernels must produce. Multi-indices are sorted tuples of input columns:
    () = u, (c,) = du/dx_c, (c, c) = d2u/dx_c2. Directions needing a second derivative come first. """
    def __init__(self, requested):
        firsts, seconds = set(), set()
        for alpha in requested:
            if len(alpha) == 1:
                firsts.add(alpha[0])
            elif len(alpha) == 2 and alpha[0] == alpha[1]:
                seconds.add(alpha[0]); firsts.add(alpha[0])
            elif len(alpha) > 0:
                raise NotImplementedError(
                    f'derivative multi-index {alpha}: the HIP kernels provide pure first and second derivatives '
                    '(mixed and third-order streams are listed under "next" in DESIGN.md)')
        self.dir_cols = sorted(seconds) + sorted(firsts - seconds)
        self.n2 = len(seconds)
        self.nd = len(self.dir_cols)
        if self.nd > MAX_DIRS:
            raise NotImplementedError(f'{self.nd} differentiated variables > {MAX_DIRS} supported by the kernels')
        self.n_streams = 1 + self.nd + self.n2
        self.index = {(): 0}
        for k, c in enumerate(self.dir_cols):
            self.index[(c,)] = 1 + k
            if k < self.n2:
                self.index[(c, c)] = 1 + self.nd + k

    def __repr__(self):
        return f'StreamSpec(dir_cols={self.dir_cols}, n2={self.n2})'


class StreamContext:
    """ Live streams of one equation evaluation; `D` asks it for derivatives (see tokens.D). """
    def __init__(self, n_inputs):
        self.n_inputs = n_inputs
        self.requested = set()
        self.tensors = {}          # alpha -> tensor [N,1]
        self.spec = None
        self.discovering = False
        self.used_autograd_fallback = False

    def tag(self, tensor, alpha):
        tensor._pinn_alpha = alpha
        tensor._pinn_ctx = self
        self.tensors[alpha] = tensor
        return tensor

    def derivative(self, alpha, col):
        new = tuple(sorted(alpha + (col,)))
        self.requested.add(new)
        if new in self.tensors:
            return self.tensors[new]
        if self.discovering:
            like = self.tensors[()]
            return self.tag(torch.rand_like(like).requires_grad_(), new)
        raise NotImplementedError(f'stream {new} was not announced by the trace of this equation')

    def stream_tensors(self):
        return list(self.tensors.items())


def discover(equation, ctx_run, n_inputs, device='cpu'):
    """ fake run (reference model_torch.py:319-325): which streams does the equation request?
    Returns (StreamSpec, needs_x_grad). """
    sc = StreamContext(n_inputs)
    sc.discovering = True
    xs = []
    for c in range(n_inputs):
        x = torch.rand((3, 1), device=device).requires_grad_()
        x._pinn_col = c
        xs.append(x)
    u = sc.tag(torch.rand((3, 1), device=device).requires_grad_(), ())
    token = active_streams.set(sc)
    try:
        ctx_run(equation, u, *xs)
    finally:
        active_streams.reset(token)
    return StreamSpec(sc.requested), sc.used_autograd_fallback


# ---------------------------------------------------------------------------------------------------------------
# symbolic tracing
# ---------------------------------------------------------------------------------------------------------------
_UNARY_TORCH = {'sin': 'SIN', 'cos': 'COS', 'exp': 'EXP', 'log': 'LOG', 'tanh': 'TANH', 'sqrt': 'SQRT', 'abs': 'ABS',
                'sigmoid': 'SIGMOID', 'neg': 'NEG', 'negative': 'NEG', 'reciprocal': 'RECIP'}
_BINARY_TORCH = {'add': 'ADD', 'sub': 'SUB', 'subtract': 'SUB', 'mul': 'MUL', 'multiply': 'MUL', 'div': 'DIV',
                 'divide': 'DIV', 'true_divide': 'DIV'}
_FOLD = {'ADD': lambda a, b: a + b, 'SUB': lambda a, b: a - b, 'MUL': lambda a, b: a * b, 'DIV': lambda a, b: a / b,
         'NEG': lambda a: -a, 'SIN': math.sin, 'COS': math.cos, 'EXP': math.exp, 'LOG': math.log, 'TANH': math.tanh,
         'SQRT': math.sqrt, 'ABS': abs, 'SIGMOID': lambda a: 1.0 / (1.0 + math.exp(-a)), 'RECIP': lambda a: 1.0 / a}


def _as_const(value):
    if isinstance(value, Sym):
        return None
    if isinstance(value, torch.nn.Parameter):
        raise TraceUnsupported('trainable variable inside the equation')
    if isinstance(value, (numbers.Real, np.floating, np.integer)) and not isinstance(value, bool):
        return float(value)
    if isinstance(value, np.ndarray) and value.size == 1:
        return float(value.reshape(()))
    if isinstance(value, torch.Tensor) and value.numel() == 1 and not value.requires_grad:
        return float(value.detach().reshape(()))
    raise TraceUnsupported(f'operand of type {type(value).__name__} cannot enter a residual program')


class Sym:
    """ node of the traced expression DAG. kind: 'stream' (alpha), 'input' (col), 'const' (value), 'op'. """
    __array_priority__ = 1000
    __array_ufunc__ = None

    def __init__(self, kind, op=None, args=(), value=None, alpha=None, col=None):
        self.kind, self.op, self.args, self.value, self.alpha, self.col = kind, op, tuple(args), value, alpha, col

    # -- construction helpers ------------------------------------------------------------------------------------
    @staticmethod
    def wrap(value):
        return value if isinstance(value, Sym) else Sym('const', value=_as_const(value))

    @staticmethod
    def make(op, *args):
        args = [Sym.wrap(a) for a in args]
        if all(a.kind == 'const' for a in args):
            return Sym('const', value=_FOLD[op](*[a.value for a in args]))
        return Sym('op', op=op, args=args)

    def __add__(self, o): return Sym.make('ADD', self, o)
    def __radd__(self, o): return Sym.make('ADD', o, self)
    def __sub__(self, o): return Sym.make('SUB', self, o)
    def __rsub__(self, o): return Sym.make('SUB', o, self)
    def __mul__(self, o): return Sym.make('MUL', self, o)
    def __rmul__(self, o): return Sym.make('MUL', o, self)
    def __truediv__(self, o): return Sym.make('DIV', self, o)
    def __rtruediv__(self, o): return Sym.make('DIV', o, self)
    def __neg__(self): return Sym.make('NEG', self)
    def __pos__(self): return self

    def __pow__(self, o):
        e = _as_const(o)
        if e is None:
            raise TraceUnsupported('power with a non-constant exponent')
        if e == 2.0:
            return Sym.make('MUL', self, self)
        if e == 1.0:
            return self
        return Sym('op', op='POW', args=(self,), value=e)

    def __rpow__(self, o):
        base = _as_const(o)
        return Sym.make('EXP', Sym.make('MUL', self, math.log(base)))

    # -- torch.* interception ------------------------------------------------------------------------------------
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        name = getattr(func, '__name__', str(func))
        kwargs = kwargs or {}
        if name in _UNARY_TORCH and len(args) == 1 and not kwargs:
            return Sym.make(_UNARY_TORCH[name], args[0])
        if name in _BINARY_TORCH and len(args) == 2 and not (set(kwargs) - {'alpha'}):
            if kwargs.get('alpha', 1) != 1:
                raise TraceUnsupported('alpha= in torch.add/sub')
            return Sym.make(_BINARY_TORCH[name], args[0], args[1])
        if name == 'pow' and len(args) == 2:
            return args[0] ** args[1] if isinstance(args[0], Sym) else Sym.wrap(args[0]).__rpow__(args[1])
        if name == 'square' and len(args) == 1:
            return Sym.make('MUL', args[0], args[0])
        if name in ('zeros_like', 'ones_like') and len(args) == 1:
            return Sym('const', value=0.0 if name == 'zeros_like' else 1.0)
        raise TraceUnsupported(f'torch.{name} is not expressible in a residual program')

    def __getattr__(self, name):
        # tensor-method spellings: x.sin(), x.pow(2), ...
        if name.startswith('__'):
            raise AttributeError(name)
        if name in _UNARY_TORCH:
            return lambda: Sym.make(_UNARY_TORCH[name], self)
        if name == 'pow':
            return lambda e: self ** e
        if name == 'square':
            return lambda: Sym.make('MUL', self, self)
        raise TraceUnsupported(f'tensor attribute .{name} is not expressible in a residual program')


def sym_D(y, x):
    """ `D` on symbolic operands: only d(stream)/d(input column). """
    if isinstance(y, Sym) and y.kind == 'stream' and isinstance(x, Sym) and x.kind == 'input':
        return Sym('stream', alpha=tuple(sorted(y.alpha + (x.col,))))
    raise TraceUnsupported('D of a composite expression')


def symbolic(equation, ctx_run, n_inputs):
    """ -> root Sym of the residual. Raises TraceUnsupported. """
    u = Sym('stream', alpha=())
    xs = [Sym('input', col=c) for c in range(n_inputs)]
    try:
        root = ctx_run(equation, u, *xs)
    except TraceUnsupported:
        raise
    except (TypeError, ValueError, AttributeError, RuntimeError, LookupError) as err:
        raise TraceUnsupported(f'{type(err).__name__}: {err}') from err
    return Sym.wrap(root)


def compile_program(root, spec, n_inputs):
    """ DAG -> (code [(op, dst, a, b)], consts [float]) in the register convention of include/pinn.h:
    registers 0..S-1 = streams, S..S+d-1 = inputs, then one fresh register per instruction. """
    S = spec.n_streams
    code, consts, memo = [], [], {}
    const_index = {}

    def const_slot(v):
        key = float(np.float32(v))
        if key not in const_index:
            const_index[key] = len(consts)
            consts.append(key)
        return const_index[key]

    def emit(op, a=0, b=0):
        dst = S + n_inputs + len(code)
        if dst >= MAX_REGS or len(code) >= MAX_OPS:
            raise TraceUnsupported('residual program too long')
        code.append((OPS[op], dst, a, b))
        return dst

    def visit(node):
        key = id(node)
        if key in memo:
            return memo[key]
        if node.kind == 'stream':
            if node.alpha not in spec.index:
                raise TraceUnsupported(f'stream {node.alpha} not in {spec}')
            reg = spec.index[node.alpha]
        elif node.kind == 'input':
            reg = S + node.col
        elif node.kind == 'const':
            reg = emit('CONST', const_slot(node.value))
        elif node.op == 'POW':
            reg = emit('POW', visit(node.args[0]), const_slot(node.value))
        elif len(node.args) == 1:
            reg = emit(node.op, visit(node.args[0]))
        else:
            ra = visit(node.args[0])
            rb = visit(node.args[1])
            reg = emit(node.op, ra, rb)
        memo[key] = reg
        return reg

    res = visit(root)
    if not code or code[-1][1] != res:
        emit('COPY', res)                       # the residual must be the value of the last instruction
    if len(consts) > MAX_CONSTS:
        raise TraceUnsupported('too many constants')
    return code, consts


def run_program_numpy(code, consts, streams, xs):
    """ fp64 host interpreter of a residual program (validation of the trace; never on the step path). """
    inv = {v: k for k, v in OPS.items()}
    S, d = streams.shape[0], xs.shape[1]
    regs = {s: streams[s].astype(np.float64) for s in range(S)}
    for c in range(d):
        regs[S + c] = xs[:, c].astype(np.float64)
    out = None
    for op, dst, a, b in code:
        name = inv[op]
        if name == 'CONST':
            out = np.full(xs.shape[0], consts[a], dtype=np.float64)
        elif name in ('ADD', 'SUB', 'MUL', 'DIV'):
            out = {'ADD': np.add, 'SUB': np.subtract, 'MUL': np.multiply, 'DIV': np.divide}[name](regs[a], regs[b])
        elif name == 'POW':
            out = np.power(regs[a], consts[b])
        elif name == 'COPY':
            out = regs[a].copy()
        else:
            fn = {'NEG': np.negative, 'SIN': np.sin, 'COS': np.cos, 'EXP': np.exp, 'LOG': np.log, 'TANH': np.tanh,
                  'SQRT': np.sqrt, 'ABS': np.abs, 'SIGMOID': lambda v: 1 / (1 + np.exp(-v)),
                  'RECIP': np.reciprocal}[name]
            out = fn(regs[a])
        regs[dst] = out
    return out
