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

from .engine import OPS, MAX_OPS, MAX_CONSTS, MAX_REGS, MAX_DIRS, MAX_AUX, MAX_VARS, RES_AFFINE, RES_PROGRAM

active_streams = ContextVar('pinn_active_streams', default=None)
# during `symbolic`: maps a trainable V(...) parameter to its user slot (or None), see Solver._variable_slot
_variable_slot = ContextVar('pinn_variable_slot', default=None)


class TraceUnsupported(Exception):
    """ the equation cannot be lowered to a residual program (generic path is used instead). """


# ---------------------------------------------------------------------------------------------------------------
# stream bookkeeping
# ---------------------------------------------------------------------------------------------------------------
def dir_code(direction):
    """ ABI code of a differentiation direction: column a, the diagonal e_a + e_b as a | (b + 1) << 4, or -- a third entry -1 --
    the minus diagonal e_a - e_b with PINN_DIR_MINUS (0x100) on top, or -- a fourth entry 2 (round 6) -- the weighted diagonal
    2 e_a +- e_b with PINN_DIR_DOUBLE (0x200), or -- six entries (a, b, sb, 1, c, sc) -- the three-column direction e_a +- e_b +- e_c with
    (c + 1) << 10 and PINN_DIR_MINUS_C (0x4000) (include/pinn.h). """
    if len(direction) == 1:
        return direction[0]
    code = (direction[0] | ((direction[1] + 1) << 4) | (0x100 if len(direction) > 2 and direction[2] < 0 else 0)
            | (0x200 if len(direction) > 3 and direction[3] == 2 else 0))
    if len(direction) > 5:
        code |= ((direction[4] + 1) << 10) | (0x4000 if direction[5] < 0 else 0)
    return code


def dir_weights(direction):
    """ [(column, weight)] of a direction tuple: (c,), (a, b) = e_a + e_b, (a, b, -1) = e_a - e_b, (a, b, +-1, 2) = 2 e_a +- e_b,
    (a, b, sb, 1, c, sc) = e_a + sb e_b + sc e_c """
    if len(direction) == 1:
        return [(direction[0], 1.0)]
    out = [(direction[0], 2.0 if len(direction) > 3 and direction[3] == 2 else 1.0),
           (direction[1], -1.0 if len(direction) > 2 and direction[2] < 0 else 1.0)]
    if len(direction) > 5:
        out.append((direction[4], -1.0 if direction[5] < 0 else 1.0))
    return out


class StreamSpec:
    """ Which derivative streams the kernels must produce. Multi-indices are sorted tuples of input columns:
    () = u, (c,) = du/dx_c, (c, c) = d2u/dx_c2, (a, b) = mixed partial. The kernels differentiate along
    *directions*: an input column, or -- for a mixed partial -- the diagonal e_a + e_b, whose second derivative
    u_vv = u_aa + 2 u_ab + u_bb yields u_ab = (u_vv - u_aa - u_bb) / 2 (stream index keyed ('d', a, b)).
    Directions needing a second derivative come first.

    One kernel call carries (nd, n2) with nd <= 3 (n2 <= nd; (3, 3) not at width 256), nd = 4 first derivatives, or -- for
    nd <= 4 -- ONE combined second-order stream (affine residuals, `combine_second_order`). Anything beyond that
    (five directions, four separate second derivatives, ...) is served by several calls over `groups` of at most two
    directions each on the generic path: the streams of different directions are independent given the network, and the
    parameter gradient is linear in the upstream stream gradients, so forward and backward split by direction. """
    def __init__(self, requested, hp=None, allact=False):
        """ allact: the net runs on the SECOND set of full breadth kernels (activation codes above 7 or nested skips: pinn_inst.inc
        PINN_ALLACT_SHAPES), which is built for the stream shapes of one or two directions per call (+ the combined second-order
        stream over two or three directions): anything larger goes through direction groups on the generic path. """
        firsts, seconds, mixed, thirds, fourths = set(), set(), set(), set(), set()
        pairs3, mixed3 = set(), {}                   # mixed THIRD-order partials (round 5): column pairs, alpha -> (pair, doubled column)
        pairs4 = set()                               # mixed FOURTH-order partials u_aabb: column pairs
        pairs31, mixed31 = set(), {}                 # u_aaab / u_abbb (round 6): column pairs, alpha -> (pair, tripled column)
        triples = set()                              # u_abc (round 6): partials of three different columns
        for alpha in requested:
            if len(alpha) == 1:
                firsts.add(alpha[0])
            elif len(alpha) == 2 and alpha[0] == alpha[1]:
                seconds.add(alpha[0])
            elif len(alpha) == 2:
                mixed.add(tuple(alpha)); seconds.update(alpha)
            elif len(alpha) == 3 and alpha[0] == alpha[1] == alpha[2]:
                thirds.add(alpha[0]); seconds.add(alpha[0])
            elif len(alpha) == 3 and len(set(alpha)) == 3:
                # u_abc = [D3_{+,+} - D3_{+,-} - D3_{-,+} + D3_{-,-}] / 24: third derivatives along e_a +- e_b +- e_c (the terms odd in b AND in c)
                triples.add(tuple(sorted(alpha)))
            elif len(alpha) == 3 and len(set(alpha)) == 2:
                # u_aab = (D3_{a+b} - D3_{a-b} - 2 u_bbb) / 6,  u_abb = (D3_{a+b} + D3_{a-b} - 2 u_aaa) / 6: third derivatives along both
                # diagonals of the pair and along the column that occurs ONCE
                a, b = sorted(set(alpha))
                doubled = a if alpha.count(a) == 2 else b
                single = b if doubled == a else a
                pairs3.add((a, b)); mixed3[tuple(alpha)] = ((a, b), doubled)
                thirds.add(single); seconds.add(single)
            elif len(alpha) == 4 and len(set(alpha)) == 1:
                fourths.add(alpha[0]); thirds.add(alpha[0]); seconds.add(alpha[0])
            elif len(alpha) == 4 and len(set(alpha)) == 2 and alpha.count(alpha[0]) == 2:
                # u_aabb = (D4_{a+b} + D4_{a-b} - 2 u_aaaa - 2 u_bbbb) / 12: fourth derivatives along both diagonals and both columns
                a, b = sorted(set(alpha))
                pairs4.add((a, b))
                for c in (a, b):
                    fourths.add(c); thirds.add(c); seconds.add(c)
            elif len(alpha) == 4 and len(set(alpha)) == 2:
                # u_aaab = (B - 2 A) / 48, u_abbb = (8 A - B) / 48 with A = D4_{a+b} - D4_{a-b}, B = D4_{2a+b} - D4_{2a-b}: fourth
                # derivatives along both diagonals and both WEIGHTED diagonals of the pair (round 6; no pure fourth derivative needed)
                a, b = sorted(set(alpha))
                pairs31.add((a, b)); mixed31[tuple(alpha)] = ((a, b), a if alpha.count(a) == 3 else b)
            elif len(alpha) > 2:
                raise NotImplementedError(
                    f'derivative multi-index {alpha}: the HIP kernels provide derivatives up to fourth order along single columns, mixed second- '
                    'and third-order partials of two columns (u_xy, u_xxy) and the mixed fourth-order ones of two columns (u_xxyy, u_xxxy, '
                    'u_xyyy) and the third-order partial of three columns (u_xyz); fourth-order partials of three or more different columns and '
                    'orders above four are not built')
        firsts |= seconds
        pairs3 |= pairs4                             # (a pair with a fourth-order diagonal carries the third-order ones anyway)
        # directions: fourth-order ones first, then the third-order ones (columns, then both diagonals of every pair a mixed third- /
        # fourth-order partial needs), then the other second-order columns, the remaining diagonals, the first-order rest
        def diag_pair(ab):
            return [ab, ab + (-1,)]
        diag4 = pairs4 | pairs31                    # pairs whose two diagonals carry a fourth derivative
        d4_dirs = ([(c,) for c in sorted(fourths)] + [d for ab in sorted(diag4) for d in diag_pair(ab)]
                   + [ab + (sign, 2) for ab in sorted(pairs31) for sign in (1, -1)])
        d3_dirs = ([(c,) for c in sorted(thirds - fourths)] + [d for ab in sorted(pairs3 - diag4) for d in diag_pair(ab)]
                   + [(a, b, sb, 1, c, sc) for a, b, c in sorted(triples) for sb in (1, -1) for sc in (1, -1)])
        self.dirs = (d4_dirs + d3_dirs + [(c,) for c in sorted(seconds - thirds)]
                     + [ab for ab in sorted(mixed) if ab not in (pairs3 | diag4)] + [(c,) for c in sorted(firsts - seconds)])
        self.n4 = len(d4_dirs)
        self.n3 = self.n4 + len(d3_dirs)
        self.n2 = self.n3 + len(seconds - thirds) + len([ab for ab in mixed if ab not in (pairs3 | diag4)])
        self.nd = len(self.dirs)
        self.n2p = self.n2 | (self.n3 << 3) | (self.n4 << 6)      # packed count of the C-ABI (include/pinn.h); only meaningful per group when large
        self.dir_cols = [dir_code(d) for d in self.dirs]
        self.n_streams = 1 + self.nd + self.n2 + self.n3 + self.n4
        self.index = {(): 0}
        for k, d in enumerate(self.dirs):
            base2, base3, base4 = 1 + self.nd + k, 1 + self.nd + self.n2 + k, 1 + self.nd + self.n2 + self.n3 + k
            if len(d) == 1:
                self.index[(d[0],)] = 1 + k
                if k < self.n2:
                    self.index[(d[0],) * 2] = base2
                if k < self.n3:
                    self.index[(d[0],) * 3] = base3
                if k < self.n4:
                    self.index[(d[0],) * 4] = base4
            elif len(d) > 5:                         # three-column direction e_a +- e_b +- e_c: its third derivative
                self.index[('d3t', d[0], d[1], d[4], d[2], d[5])] = base3
            elif len(d) > 3:                         # weighted diagonal 2 e_a +- e_b: only its fourth derivative is asked for
                self.index[('d4w',) + d[:3]] = base4
            else:
                sign = -1 if len(d) > 2 else 1
                if sign > 0:
                    self.index[('d',) + d[:2]] = base2
                if k < self.n3:
                    self.index[('d3',) + d[:2] + (sign,)] = base3
                if k < self.n4:
                    self.index[('d4',) + d[:2] + (sign,)] = base4
        self.mixed = {ab: (self.index[('d',) + ab], self.index[(ab[0], ab[0])], self.index[(ab[1], ab[1])])
                      for ab in sorted(mixed)}
        # alpha -> (stream of D3 along a + b, of D3 along a - b, of the pure third derivative it subtracts, sign of the minus-diagonal term)
        self.mixed3 = {}
        for alpha, ((a, b), doubled) in sorted(mixed3.items()):
            single = b if doubled == a else a
            self.mixed3[alpha] = (self.index[('d3', a, b, 1)], self.index[('d3', a, b, -1)], self.index[(single,) * 3],
                                  -1.0 if doubled == a else 1.0)
        # (a, a, b, b) -> streams of D4 along a + b, a - b, u_aaaa, u_bbbb
        self.mixed4 = {(a, a, b, b): (self.index[('d4', a, b, 1)], self.index[('d4', a, b, -1)], self.index[(a,) * 4], self.index[(b,) * 4])
                       for a, b in sorted(pairs4)}
        # (a, a, a, b) / (a, b, b, b) -> [(stream, coefficient)]: streams of D4 along a + b, a - b, 2a + b, 2a - b
        self.mixed31 = {}
        for alpha, ((a, b), tripled) in sorted(mixed31.items()):
            ip, im = self.index[('d4', a, b, 1)], self.index[('d4', a, b, -1)]
            i2p, i2m = self.index[('d4w', a, b, 1)], self.index[('d4w', a, b, -1)]
            if tripled == a:
                self.mixed31[alpha] = [(i2p, 1.0 / 48.0), (i2m, -1.0 / 48.0), (ip, -2.0 / 48.0), (im, 2.0 / 48.0)]
            else:
                self.mixed31[alpha] = [(ip, 8.0 / 48.0), (im, -8.0 / 48.0), (i2p, -1.0 / 48.0), (i2m, 1.0 / 48.0)]
        # (a, b, c) -> [(stream, coefficient)]: third derivatives along e_a +- e_b +- e_c
        self.mixed111 = {abc: [(self.index[('d3t',) + abc + (sb, sc)], sb * sc / 24.0) for sb in (1, -1) for sc in (1, -1)] for abc in sorted(triples)}
        # can ONE kernel call produce all of it as separate streams?
        if self.n4 > 0:
            # fourth order in ONE call: that direction alone (u'''' = f(x) beams, u_t-free fourth-order ODEs); anything else in groups
            self.single_call = self.n4 == 1 and self.nd == 1
        elif self.n3 > 0:
            # third order in ONE call: one such direction, at most two directions in all, nothing else of second order (u_xxx-type
            # equations: KdV in (x, t), third-order ODEs). Anything else with third derivatives (two third-order columns, a
            # third-order column beside other second-order ones, the diagonals of a mixed third-order partial) goes through the groups below
            self.single_call = self.n3 == 1 and self.n2 == 1 and self.nd <= 2
        else:
            self.single_call = (self.nd <= 3 and not (self.nd == 3 and self.n2 == 3 and hp == 256)) or \
                               (self.nd == 4 and self.n2 == 0)
        if allact and self.n4 == 0:
            self.single_call = (self.nd <= 2) if self.n3 == 0 else (self.n3 == 1 and self.n2 == 1 and self.nd == 1)
        self.hp = hp
        if hp == 512:
            # width 512 (round 6): one 16-point tile of S streams is 33 KB x S of LDS activations -- S <= 3 per kernel call: the value with ONE
            # direction and its second derivative, or with two first-order directions; everything else in direction groups of that size
            if self.n3 > 0:
                raise NotImplementedError('third- and fourth-order derivatives at hidden width 512 are not built (three derivative streams per '
                                          'kernel call at that width); widths up to 256 carry them')
            if allact:
                raise NotImplementedError('hidden width 512: the second set of full breadth kernels (activation codes above 7, nested skips) is '
                                          'not built at that width')
            self.single_call = self.nd <= 1 or (self.nd == 2 and self.n2 == 0)
        # ... or as [u, firsts, one combined second-order stream] (affine residuals only)
        self.combinable = 2 <= self.nd <= (3 if allact else MAX_DIRS) and self.n2 >= 1 and self.n3 == 0 and hp != 512
        # groups for the generic path: (direction codes, packed n2 of the group, stream index of each of the group's streams)
        self.groups = []
        if self.single_call:
            chunks = [list(range(self.nd))] if self.nd else [[]]
        else:
            # every third- / fourth-order direction in a call of its own (those kernels carry one such direction), the rest in pairs
            rest = list(range(self.n3, self.nd))
            chunks = [[k] for k in range(self.n3)] + [rest[i:i + 2] for i in range(0, len(rest), 2)]
            if hp == 512:       # second-order directions one per call, the first-order rest in pairs
                rest = list(range(self.n2, self.nd))
                chunks = [[k] for k in range(self.n2)] + [rest[i:i + 2] for i in range(0, len(rest), 2)]
        for ks in chunks:
            n2g = sum(1 for k in ks if k < self.n2)
            n3g = sum(1 for k in ks if k < self.n3)
            n4g = sum(1 for k in ks if k < self.n4)
            idx = ([0] + [1 + k for k in ks] + [1 + self.nd + k for k in ks if k < self.n2]
                   + [1 + self.nd + self.n2 + k for k in ks if k < self.n3]
                   + [1 + self.nd + self.n2 + self.n3 + k for k in ks if k < self.n4])
            self.groups.append(([self.dir_cols[k] for k in ks], n2g | (n3g << 3) | (n4g << 6), idx))

    def __repr__(self):
        return f'StreamSpec(dirs={self.dirs}, n2={self.n2})'


class StreamContext:
    """ Live streams of one equation evaluation; `D` asks it for derivatives (see tokens.D). """
    def __init__(self, n_inputs):
        self.n_inputs = n_inputs
        self.requested = set()
        self.tensors = {}          # alpha -> tensor [N,1]
        self.spec = None
        self.discovering = False
        self.used_autograd_fallback = False
        # a custom forward() that hands `self.conv_block` a (non-affine or column-mixing) map ys = phi(points) of the batch: the streams are
        # derivatives with respect to the NETWORK's input columns, taken at ys; a derivative with respect to a column of the solver is
        # sum_k stream_{alpha + k} * d ys_k / d x_col (chain rule; the Jacobian of phi by torch autograd on the user's few pointwise ops)
        self.ymap = None           # {'cols': the solver's column tensors, 'pattern': [k][col] "ys_k depends on x_col", 'ys': None, 'jac': {}}

    def tag(self, tensor, alpha):
        tensor._pinn_alpha = alpha
        tensor._pinn_ctx = self
        self.tensors[alpha] = tensor
        return tensor

    def derivative(self, alpha, col):
        if self.ymap is not None:
            return self._through_map(alpha, col)
        return self._stream(alpha, col)

    def _through_map(self, alpha, col):
        m = self.ymap
        if m['ys'] is None:
            raise NotImplementedError('D(...) of the network before the forward() of the model has called self.conv_block')
        total = None
        for k, row in enumerate(m['pattern']):
            if not row[col]:
                continue
            if (k, col) not in m['jac']:
                m['jac'][(k, col)] = torch.autograd.grad(m['ys'][:, k].sum(), m['cols'][col], retain_graph=True, create_graph=True)[0]
            term = m['jac'][(k, col)] * self._stream(alpha, k)
            total = term if total is None else total + term
        return total if total is not None else torch.zeros_like(self.tensors[()])

    def _stream(self, alpha, col):
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


def call_with_streams(sc, equation, *args):
    """ run the user's callable with `sc` as the active stream context. Must itself be what `ctx.run` invokes: a
    ContextVar set outside is not visible inside the solver's own contextvars.Context (model_torch.py:316-317). """
    token = active_streams.set(sc)
    try:
        return equation(*args)
    finally:
        active_streams.reset(token)


def discover(equation, ctx_run, n_inputs, device='cpu', hp=None, allact=False):
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
    ctx_run(call_with_streams, sc, equation, u, *xs)
    return StreamSpec(sc.requested, hp=hp, allact=allact), sc.used_autograd_fallback


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
        lookup = _variable_slot.get()
        slot = lookup(value) if lookup is not None else None
        if slot is None or slot >= MAX_VARS:
            raise TraceUnsupported('trainable variable inside the equation that is not one of the first '
                                   f'{MAX_VARS} scalar V(...) slots of the model')
        return None
    if isinstance(value, (numbers.Real, np.floating, np.integer)) and not isinstance(value, bool):
        return float(value)
    if isinstance(value, np.ndarray) and value.size == 1:
        return float(value.reshape(()))
    if isinstance(value, torch.Tensor) and value.numel() == 1 and not value.requires_grad:
        return float(value.detach().reshape(()))
    raise TraceUnsupported(f'operand of type {type(value).__name__} cannot enter a residual program')


class Sym:
    """ node of the traced expression DAG. kind: 'stream' (alpha), 'input' (col), 'const' (value), 'op',
    'var' (col = user slot of a trainable scalar V(...)). """
    __array_priority__ = 1000
    __array_ufunc__ = None

    def __init__(self, kind, op=None, args=(), value=None, alpha=None, col=None):
        self.kind, self.op, self.args, self.value, self.alpha, self.col = kind, op, tuple(args), value, alpha, col

    # -- construction helpers ------------------------------------------------------------------------------------
    @staticmethod
    def wrap(value):
        if isinstance(value, Sym):
            return value
        const = _as_const(value)
        if const is None:                       # a scalar trainable V(...): lives in a program register of its own
            return Sym('var', col=_variable_slot.get()(value))
        return Sym('const', value=const)

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
        e = None if isinstance(o, torch.nn.Parameter) else _as_const(o)
        if e is None:
            raise TraceUnsupported('power with a non-constant exponent')
        if e == 2.0:
            return Sym.make('MUL', self, self)
        if e == 1.0:
            return self
        return Sym('op', op='POW', args=(self,), value=e)

    def __rpow__(self, o):
        base = None if isinstance(o, torch.nn.Parameter) else _as_const(o)
        if base is None:
            raise TraceUnsupported('power with a non-constant base')
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


def _is_const(node, value=None):
    return node.kind == 'const' and (value is None or node.value == value)


def _s_add(a, b):
    return b if _is_const(a, 0.0) else a if _is_const(b, 0.0) else Sym.make('ADD', a, b)


def _s_sub(a, b):
    return a if _is_const(b, 0.0) else Sym.make('NEG', b) if _is_const(a, 0.0) else Sym.make('SUB', a, b)


def _s_mul(a, b):
    if _is_const(a, 0.0) or _is_const(b, 0.0):
        return Sym('const', value=0.0)
    return b if _is_const(a, 1.0) else a if _is_const(b, 1.0) else Sym.make('MUL', a, b)


def _keep(memo, node):
    """ the memo dictionaries of this module are keyed by id(node): every memoised node is kept alive beside its entry, or a node built
    later could be handed the id -- and the memoised answer -- of one that is gone (round 5: that is how the emitter lost a callable IC) """
    memo.setdefault('_alive', []).append(node)


def _differentiate(node, col, memo):
    """ symbolic d(node)/d(input column `col`): streams step to the next derivative stream (`D` of the field), x-only
    sub-expressions differentiate in closed form, everything else by the chain rule. """
    key = id(node)
    if key in memo:
        return memo[key]
    zero, one = Sym('const', value=0.0), Sym('const', value=1.0)
    if node.kind in ('const', 'var'):
        out = zero
    elif node.kind == 'input':
        out = one if node.col == col else zero
    elif node.kind == 'stream':
        if not all(isinstance(c, int) for c in node.alpha) or len(node.alpha) >= 3 or \
                (len(node.alpha) == 2 and not (node.alpha[0] == node.alpha[1] == col)):
            raise TraceUnsupported('derivative of a second-order stream beyond u_ccc (mixed third order / fourth order)')
        alpha = tuple(sorted(node.alpha + (col,)))
        if len(alpha) == 2 and alpha[0] != alpha[1]:
            # mixed partial by polarisation over the diagonal direction e_a + e_b
            a, b = alpha
            out = (Sym('stream', alpha=('d', a, b)) - Sym('stream', alpha=(a, a)) - Sym('stream', alpha=(b, b))) * 0.5
        else:
            out = Sym('stream', alpha=alpha)
    else:
        op, args = node.op, node.args
        d = [_differentiate(a, col, memo) for a in args]
        a = args[0]
        if op == 'ADD':
            out = _s_add(d[0], d[1])
        elif op == 'SUB':
            out = _s_sub(d[0], d[1])
        elif op == 'MUL':
            out = _s_add(_s_mul(d[0], args[1]), _s_mul(a, d[1]))
        elif op == 'DIV':
            b = args[1]
            out = _s_sub(Sym.make('DIV', d[0], b) if not _is_const(d[0], 0.0) else zero,
                         _s_mul(Sym.make('DIV', a, Sym.make('MUL', b, b)), d[1]))
        elif op in ('NEG', 'COPY'):
            out = Sym.make('NEG', d[0]) if op == 'NEG' and not _is_const(d[0], 0.0) else d[0]
        elif _is_const(d[0], 0.0):
            out = zero
        elif op == 'SIN':
            out = _s_mul(Sym.make('COS', a), d[0])
        elif op == 'COS':
            out = _s_mul(Sym.make('NEG', Sym.make('SIN', a)), d[0])
        elif op == 'EXP':
            out = _s_mul(node, d[0])
        elif op == 'LOG':
            out = Sym.make('DIV', d[0], a)
        elif op == 'TANH':
            out = _s_mul(Sym.make('SUB', 1.0, Sym.make('MUL', node, node)), d[0])
        elif op == 'SIGMOID':
            out = _s_mul(Sym.make('MUL', node, Sym.make('SUB', 1.0, node)), d[0])
        elif op == 'SQRT':
            out = Sym.make('DIV', d[0], Sym.make('MUL', 2.0, node))
        elif op == 'RECIP':
            out = Sym.make('NEG', _s_mul(Sym.make('MUL', node, node), d[0]))
        elif op == 'POW':
            out = _s_mul(Sym.make('MUL', node.value, a ** (node.value - 1.0)), d[0])
        else:
            raise TraceUnsupported(f'D through {op}')
    memo[key] = out
    _keep(memo, node)
    return out


def sym_D(y, x):
    """ `D` on symbolic operands: d(stream)/d(input column) is the next stream; composite expressions -- D(f * f, x),
    D(a(x) * D(f, x), x) -- are differentiated symbolically (chain rule down to the streams). """
    if not (isinstance(x, Sym) and x.kind == 'input'):
        raise TraceUnsupported('D with respect to something that is not an input column')
    return _differentiate(Sym.wrap(y), x.col, {})


def _call_with_variables(lookup, equation, *args):
    """ run the callable with `lookup` resolving trainable V(...) parameters (must itself be what ctx.run invokes,
    cf. call_with_streams) """
    token = _variable_slot.set(lookup)
    try:
        return Sym.wrap(equation(*args))
    finally:
        _variable_slot.reset(token)


def symbolic(equation, ctx_run, n_inputs, variable_slot=None):
    """ -> root Sym of the residual. Raises TraceUnsupported. `variable_slot(param)` -> user slot of a scalar
    trainable V(...) or None. """
    u = Sym('stream', alpha=())
    xs = [Sym('input', col=c) for c in range(n_inputs)]
    try:
        root = ctx_run(_call_with_variables, variable_slot, equation, u, *xs)
    except TraceUnsupported:
        raise
    except (TypeError, ValueError, AttributeError, RuntimeError, LookupError) as err:
        raise TraceUnsupported(f'{type(err).__name__}: {err}') from err
    return Sym.wrap(root)


def _depends_on_inputs(node, memo):
    key = id(node)
    if key not in memo:
        memo[key] = node.kind == 'input' or any(_depends_on_inputs(a, memo) for a in node.args)
        _keep(memo, node)
    return memo[key]


def symbolic_constraint(constraint, ctx_run, n_inputs, to_points, variable_slot=None):
    """ Trace a constraint callable `constraint(f, *xs)` (reference model_torch.py:451-457: `f` evaluates the model on
    whatever points it is given, e.g. `lambda f, x: f(torch.tensor([0.5]))`). Supported form: ONE call of `f` on
    constants -- numbers, arrays, tensors, cast by `to_points` exactly like the reference's `reshape_and_concat` -- and
    a pointwise expression of its value and trainable variables that does not touch the batch columns.
    Returns (root Sym over the value stream, points [n_c, n_inputs] float32). Raises TraceUnsupported otherwise. """
    calls = []

    def model_at(*args):
        if any(isinstance(a, Sym) for a in args):
            raise TraceUnsupported('constraint evaluates the model on the batch points')
        if calls:
            raise TraceUnsupported('constraint evaluates the model more than once')
        pts = np.asarray(to_points(args), dtype=np.float32)
        if pts.ndim != 2 or pts.shape[1] != n_inputs:
            raise TraceUnsupported(f'constraint calls the model with {pts.shape[-1]} columns, the problem has {n_inputs}')
        calls.append(pts)
        return Sym('stream', alpha=())

    xs = [Sym('input', col=c) for c in range(n_inputs)]
    try:
        root = ctx_run(_call_with_variables, variable_slot, constraint, model_at, *xs)
    except TraceUnsupported:
        raise
    except (TypeError, ValueError, AttributeError, RuntimeError, LookupError) as err:
        raise TraceUnsupported(f'{type(err).__name__}: {err}') from err
    if not calls:
        raise TraceUnsupported('constraint does not evaluate the model')
    if _depends_on_inputs(root, {}):
        raise TraceUnsupported('constraint depends on the batch points')
    return root, calls[0]


def kernel_stream_shift(nd, n2p):
    """ extra second-derivative streams of the kernel instantiation that serves (nd, n2): mirror of pick_n2 in csrc/pinn_abi.cpp (the
    compiled second-order shapes per nd; third / fourth order: exact instantiations only) """
    if n2p > 7:
        return 0
    avail = {0: (0,), 1: (0, 1), 2: (0, 2), 3: (2, 3), 4: (0,)}.get(nd, ())
    fits = [k for k in avail if k >= n2p]
    return (min(fits) - n2p) if fits else 0


class _Emitter:
    """ register-code emitter shared by the pre-pass and the main program. Main programs are single-assignment (the
    reverse sweep of the interpreter relies on it); the pre-pass has no reverse sweep, so with `reuse=True` its code is
    emitted over virtual registers and `finish()` maps them onto the MAX_REGS physical ones, recycling a register after
    its last use (an initial condition with its derivatives easily takes more ops than there are registers). """
    def __init__(self, first_temp, reuse=False, limit=MAX_REGS):
        self.first_temp, self.code, self.consts, self.memo, self._cidx = first_temp, [], [], {}, {}
        self.reuse, self.limit = reuse, limit
        # the memo is keyed by id(node): every visited node is kept alive here, or a node created later (the derivatives of a callable
        # IC, emitted after the residual's own rows) can be handed the id of a temporary that is gone -- and with it that temporary's
        # register. (Round 5: the tutorial's heat equation with a parameter column lost its lowered IC that way -- the validation against
        # the callable caught it, the fit fell back to torch autograd per iteration: 470 us instead of ~20 us per iteration)
        self._alive = []

    def const_slot(self, v):
        # (unrounded: the x-only pre-pass runs in fp64 and reads its constants as doubles -- pinn_residual_t::pre_consts64; the fp32
        #  main program gets them rounded once, when engine.Program stores them. Rounded to fp32 HERE, round 6's first fp64 pre-pass
        #  still carried half of the systematic error of fl32(pi) in e pi cos(e pi x): tools/cfg4_bl_probe.py)
        key = float(v)
        if key not in self._cidx:
            if len(self.consts) >= MAX_CONSTS:
                raise TraceUnsupported('too many constants')
            self._cidx[key] = len(self.consts)
            self.consts.append(key)
        return self._cidx[key]

    def emit(self, op, a=0, b=0):
        dst = self.first_temp + sum(1 for c in self.code if c[0] != OPS['STORE'])
        if (dst >= self.limit and not self.reuse) or len(self.code) >= MAX_OPS:
            raise TraceUnsupported('residual program too long')
        self.code.append((OPS[op], dst, a, b))
        return dst

    def visit(self, node, leaf):
        key = id(node)
        if key in self.memo:
            return self.memo[key]
        reg = leaf(node)
        if reg is None:
            if node.kind == 'const':
                reg = self.emit('CONST', self.const_slot(node.value))
            elif node.op == 'POW':
                reg = self.emit('POW', self.visit(node.args[0], leaf), self.const_slot(node.value))
            elif len(node.args) == 1:
                reg = self.emit(node.op, self.visit(node.args[0], leaf))
            else:
                ra = self.visit(node.args[0], leaf)
                rb = self.visit(node.args[1], leaf)
                reg = self.emit(node.op, ra, rb)
        self.memo[key] = reg
        self._alive.append(node)
        return reg

    def finish(self):
        """ -> (code, consts) over physical registers """
        if not self.reuse:
            return self.code, self.consts
        two_reg = {OPS[k] for k in ('ADD', 'SUB', 'MUL', 'DIV')}
        last_use = {}
        for i, (op, dst, a, b) in enumerate(self.code):
            if op == OPS['STORE']:
                last_use[a] = i
                continue
            if op != OPS['CONST']:
                last_use[a] = i
            if op in two_reg:
                last_use[b] = i
        phys, free, out = {}, [], []
        nxt = self.first_temp

        def reg(v):
            return v if v < self.first_temp else phys[v]
        for i, (op, dst, a, b) in enumerate(self.code):
            if op == OPS['STORE']:
                out.append((op, 0, reg(a), b))
                srcs = [a]
            else:
                ra = a if op == OPS['CONST'] else reg(a)
                rb = reg(b) if op in two_reg else b
                srcs = ([] if op == OPS['CONST'] else [a]) + ([b] if op in two_reg else [])
                # sources that die here may lend their register to the result
                for v in srcs:
                    if v >= self.first_temp and last_use.get(v) == i and phys[v] not in free:
                        free.append(phys[v])
                if free:
                    p = free.pop()
                else:
                    p = nxt
                    nxt += 1
                    if p >= MAX_REGS:
                        raise TraceUnsupported('pre-pass needs more than %d live registers' % MAX_REGS)
                phys[dst] = p
                out.append((op, p, ra, rb))
                continue
            for v in srcs:
                if v >= self.first_temp and last_use.get(v) == i and phys[v] not in free:
                    free.append(phys[v])
        return out, self.consts


def _uses_streams(node, memo):
    key = id(node)
    if key not in memo:
        # (expressions of trainable variables stay in the main program too: the pre-pass has no reverse sweep)
        memo[key] = node.kind in ('stream', 'var') or any(_uses_streams(a, memo) for a in node.args)
        _keep(memo, node)
    return memo[key]


def _affine(node, spec, memo):
    """ node as sum_s C_s(x) * stream_s + F(x): returns ({stream index: x-only Sym}, x-only Sym) or None. """
    key = id(node)
    if key in memo:
        return memo[key]
    out = None
    if node.kind == 'stream':
        if node.alpha not in spec.index:
            raise TraceUnsupported(f'stream {node.alpha} not in {spec}')
        out = ({spec.index[node.alpha]: Sym('const', value=1.0)}, Sym('const', value=0.0))
    elif node.kind in ('input', 'const'):
        out = ({}, node)
    elif node.kind == 'var':
        out = None                                                 # trainable coefficient: needs the program's reverse sweep
    else:
        parts = [_affine(a, spec, memo) for a in node.args]
        if all(p is not None for p in parts):
            if all(not p[0] for p in parts):
                out = ({}, node)                                   # x-only sub-expression: opaque source term
            elif node.op in ('ADD', 'SUB'):
                (ca, fa), (cb, fb) = parts
                sign = (lambda v: v) if node.op == 'ADD' else (lambda v: -v)
                coefs = dict(ca)
                for k, v in cb.items():
                    coefs[k] = coefs[k] + sign(v) if k in coefs else sign(v)
                out = (coefs, fa + sign(fb))
            elif node.op == 'NEG':
                out = ({k: -v for k, v in parts[0][0].items()}, -parts[0][1])
            elif node.op == 'MUL' and (not parts[0][0] or not parts[1][0]):
                (lin, scale) = (parts[1], parts[0][1]) if not parts[0][0] else (parts[0], parts[1][1])
                out = ({k: v * scale for k, v in lin[0].items()}, lin[1] * scale)
            elif node.op == 'DIV' and not parts[1][0]:
                out = ({k: v / parts[1][1] for k, v in parts[0][0].items()}, parts[0][1] / parts[1][1])
    memo[key] = out
    _keep(memo, node)
    return out


def _second_order_split(node, spec, memo):
    """ node as  sum_k c_k * (second-order stream k) + REST  with CONSTANT c_k and REST free of second-order streams:
    returns ({stream index: float}, REST Sym) or None (a second derivative inside a nonlinear term or under a non-constant
    factor). What a residual PROGRAM needs in order to run on ONE combined second-order stream (reaction-diffusion, nonlinear
    Poisson / Helmholtz, Allen-Cahn, viscous Burgers in several space dimensions: the Laplacian part is linear, the rest is not). """
    key = id(node)
    if key in memo:
        return memo[key]
    first2 = 1 + spec.nd
    out = None
    if node.kind == 'stream':
        if node.alpha not in spec.index:
            raise TraceUnsupported(f'stream {node.alpha} not in {spec}')
        idx = spec.index[node.alpha]
        out = ({idx: 1.0}, Sym('const', value=0.0)) if idx >= first2 else ({}, node)
    elif node.kind in ('input', 'const', 'var'):
        out = ({}, node)
    else:
        parts = [_second_order_split(a, spec, memo) for a in node.args]
        if all(p is not None for p in parts):
            if all(not p[0] for p in parts):
                out = ({}, node)                                   # no second-order stream below: opaque
            elif node.op in ('ADD', 'SUB'):
                (ca, fa), (cb, fb) = parts
                sign = 1.0 if node.op == 'ADD' else -1.0
                coefs = dict(ca)
                for k, v in cb.items():
                    coefs[k] = coefs.get(k, 0.0) + sign * v
                out = (coefs, _s_add(fa, fb) if sign > 0 else _s_sub(fa, fb))
            elif node.op == 'NEG':
                out = ({k: -v for k, v in parts[0][0].items()}, _s_sub(Sym('const', value=0.0), parts[0][1]))
            elif node.op == 'MUL' and any(a.kind == 'const' for a in node.args):
                ci = 0 if node.args[0].kind == 'const' else 1
                scale, lin = float(node.args[ci].value), parts[1 - ci]
                out = ({k: v * scale for k, v in lin[0].items()}, _s_mul(Sym('const', value=scale), lin[1]))
            elif node.op == 'DIV' and node.args[1].kind == 'const' and float(node.args[1].value) != 0.0:
                scale = 1.0 / float(node.args[1].value)
                out = ({k: v * scale for k, v in parts[0][0].items()}, _s_mul(Sym('const', value=scale), parts[0][1]))
    memo[key] = out
    _keep(memo, node)
    return out


class ResidualPlan:
    """ host-side description of a lowered residual (see pinn_residual_t in include/pinn.h). """
    def __init__(self, kind, n_inputs, n_streams, pre, n_aux, program=None, coef=None, coef_row=None, src_const=0.0,
                 src_row=-1, n_vars=0):
        self.kind, self.n_inputs, self.n_streams = kind, n_inputs, n_streams
        self.n_vars = n_vars        # trainable V(...) scalars (user slots 0..n_vars-1) the program reads as registers
        self.pre, self.n_aux, self.program = pre, n_aux, program
        self.coef = coef or [0.0] * n_streams
        self.coef_row = coef_row or [-1] * n_streams
        self.src_const, self.src_row = src_const, src_row
        self.comb_w = None          # set by `combine_second_order`: weights of the single combined second-order stream
        # set by `attach_initial_condition`: the callable IC and its derivative streams as pre-pass rows / constants
        self.ic_row, self.ic_const = None, None
        self._pre_emitter, self._rows = None, None

    @property
    def kernel_n2(self):
        """ number of second-order streams the kernel propagates for this residual """
        return 1 if self.comb_w is not None else None

    def to_struct(self):
        from .engine import Residual
        return Residual.build(self.kind, self.n_aux, self.pre if self.n_aux else None, self.program, self.coef,
                              self.coef_row, self.src_const, self.src_row, self.comb_w, n_vars=self.n_vars,
                              ic_row=self.ic_row, ic_const=self.ic_const)


def _worth_combining(weights, spec):
    nonzero = sum(1 for w in weights if w != 0.0)
    return nonzero >= 2 or (nonzero == 1 and spec.n2 < spec.nd)


def combine_second_order(plan, spec):
    # (width 512: no combined second-order stream -- S <= 3 streams per call at that width, StreamSpec)
    if getattr(spec, 'hp', None) == 512:
        return False
    return _combine_second_order(plan, spec)


def _combine_second_order(plan, spec):
    """ Affine residual whose second derivatives enter only as  sum_k c_k u_kk  with CONSTANT c_k (Laplacian, wave,
    heat operators): propagate that one combination instead of n2 separate streams. Rewrites the plan in place to the
    stream layout [u, firsts (nd), combined] and returns True; otherwise leaves it alone. """
    if plan.kind != RES_AFFINE or spec.n2 < 1 or spec.nd < 2 or spec.n3 > 0 or spec.nd > MAX_DIRS or plan.comb_w is not None:
        return False
    first2 = 1 + spec.nd
    if any(plan.coef_row[first2 + k] >= 0 for k in range(spec.n2)):
        return False                                   # x-dependent coefficient on a second derivative
    weights = [plan.coef[first2 + k] if k < spec.n2 else 0.0 for k in range(spec.nd)]
    # (ONE second derivative beside first-order directions -- u_t = nu u_xx, advection-diffusion in (x, t) -- combines as well: the
    #  kernels are built for n2 = 0 / nd second-order streams per call, so (nd, 1) would otherwise run padded to the next size)
    if not _worth_combining(weights, spec):
        return False
    plan.coef = plan.coef[:first2] + [1.0]
    plan.coef_row = plan.coef_row[:first2] + [-1]
    plan.n_streams = first2 + 1
    plan.comb_w = weights
    return True



def symbolic_initial_condition(initial_condition, ctx_run, n_spatial):
    """ the callable IC (reference model_torch.py:124-127: called with the 1-D spatial columns) as a Sym DAG over input
    columns 0 .. n_spatial-1. Raises TraceUnsupported (trainable variables, tensor constants, ops outside the program ISA). """
    xs = [Sym('input', col=c) for c in range(n_spatial)]
    try:
        root = ctx_run(_call_with_variables, None, initial_condition, *xs)
    except TraceUnsupported:
        raise
    except (TypeError, ValueError, AttributeError, RuntimeError, LookupError) as err:
        raise TraceUnsupported(f'{type(err).__name__}: {err}') from err

    def has_var(node, seen):
        if id(node) in seen:
            return False
        seen.add(id(node))
        return node.kind in ('var', 'stream') or any(has_var(a, seen) for a in node.args)
    if has_var(root, set()):
        raise TraceUnsupported('initial condition depends on trainable variables')
    return root


def _emit_ic_rows(pre, pre_leaf, rows, spec, ic_root, comb_w):
    """ IC(x) and its derivative streams as further pre-pass rows -> (ic_row, ic_const) in the kernel's stream layout """
    def along(node, direction):
        total = Sym('const', value=0.0)
        for c in direction:
            total = _s_add(total, _differentiate(node, c, {}))
        return total
    firsts = [along(ic_root, d) for d in spec.dirs]
    seconds = [along(firsts[k], spec.dirs[k]) for k in range(spec.n2)]
    thirds = [along(seconds[k], spec.dirs[k]) for k in range(spec.n3)]
    if comb_w is not None:
        comb = Sym('const', value=0.0)
        for k, w in enumerate(comb_w):
            if w != 0.0 and k < len(seconds):
                comb = _s_add(comb, _s_mul(Sym('const', value=float(w)), seconds[k]))
        exprs = [ic_root] + firsts + [comb]
    else:
        exprs = [ic_root] + firsts + seconds + thirds
    ic_row, ic_const = [-1] * len(exprs), [0.0] * len(exprs)
    for s_idx, expr in enumerate(exprs):
        if expr.kind == 'const':
            ic_const[s_idx] = expr.value
            continue
        reg = pre.visit(expr, pre_leaf)
        if reg not in rows:
            if len(rows) >= MAX_AUX:
                raise TraceUnsupported(f'more than {MAX_AUX} pre-pass rows with the initial condition')
            rows.append(reg)
            pre.code.append((OPS['STORE'], 0, reg, len(rows) - 1))
            if len(pre.code) > MAX_OPS:
                raise TraceUnsupported('pre-pass program too long with the initial condition')
        ic_row[s_idx] = rows.index(reg)
    return ic_row, ic_const


def attach_initial_condition(plan, spec, ic_root):
    """ IC(x) and the derivative streams the ansatz adds to u (model_torch.py:124-127 under nested D(...)): value, first /
    second directional derivatives along the spec's directions (the combined second-order stream when the plan has one),
    differentiated symbolically and appended to the plan's x-only pre-pass as further rows -- the step then contains no
    torch autograd over the IC callable. Streams that are constant (zero: directions the IC does not depend on) become
    plain numbers. AFFINE plans only (call it after `combine_second_order`); a residual PROGRAM numbers its registers
    behind the pre-pass rows, so there the rows are emitted by `lower_residual(..., ic_root=...)` itself.
    Raises TraceUnsupported when rows / registers / ops run out (the caller keeps torch autograd). """
    if plan._pre_emitter is None or plan.kind != RES_AFFINE:
        raise TraceUnsupported('plan without an open pre-pass emitter')
    (pre, pre_leaf), rows = plan._pre_emitter, plan._rows
    saved = (list(pre.code), list(pre.consts), dict(pre.memo), dict(pre._cidx), list(rows))
    try:
        ic_row, ic_const = _emit_ic_rows(pre, pre_leaf, rows, spec, ic_root, plan.comb_w)
        plan.pre = pre.finish()
    except TraceUnsupported:
        pre.code[:], pre.consts[:] = saved[0], saved[1]
        pre.memo.clear(); pre.memo.update(saved[2])
        pre._cidx.clear(); pre._cidx.update(saved[3])
        rows[:] = saved[4]
        raise
    plan.n_aux = len(rows)
    plan.ic_row, plan.ic_const = ic_row, ic_const
    return plan


def run_ic_numpy(plan, xs):
    """ fp64 host evaluation of the IC rows of a plan: [S_kernel, N] (validation of `attach_initial_condition`) """
    n, d = xs.shape
    aux = {}
    regs = {c: xs[:, c].astype(np.float64) for c in range(d)}
    _run_code_numpy(plan.pre[0], plan.pre[1], regs, n, aux)
    return np.stack([aux[r] if r >= 0 else np.full(n, c, dtype=np.float64) for r, c in zip(plan.ic_row, plan.ic_const)])


def lower_residual(root, spec, n_inputs, ic_root=None):
    """ DAG -> ResidualPlan. x-only sub-expressions go to the pre-pass; if the rest is affine in the streams the
    step needs no interpreter at all (every linear PDE), otherwise a register program is emitted. """
    S = spec.n_streams
    pre = _Emitter(first_temp=n_inputs, reuse=True)
    rows = []

    def pre_leaf(node):
        if node.kind == 'stream':
            raise TraceUnsupported('internal: stream inside an x-only expression')
        return node.col if node.kind == 'input' else None

    def aux_row(expr):
        reg = pre.visit(expr, pre_leaf)
        if reg not in rows:
            if len(rows) >= MAX_AUX:
                raise TraceUnsupported(f'more than {MAX_AUX} x-only sub-expressions')
            rows.append(reg)
            pre.code.append((OPS['STORE'], 0, reg, len(rows) - 1))
            if len(pre.code) > MAX_OPS:
                raise TraceUnsupported('pre-pass program too long')
        return rows.index(reg)

    aff = _affine(root, spec, {})
    if aff is not None:
        coefs, src = aff
        coef, coef_row = [0.0] * S, [-1] * S
        for idx, expr in coefs.items():
            if expr.kind == 'const':
                coef[idx] = expr.value
            else:
                coef_row[idx] = aux_row(expr)
        src_const, src_row = (src.value, -1) if src.kind == 'const' else (0.0, aux_row(src))
        plan = ResidualPlan(RES_AFFINE, n_inputs, S, pre.finish(), len(rows), coef=coef, coef_row=coef_row,
                            src_const=src_const, src_row=src_row)
        plan._pre_emitter, plan._rows = (pre, pre_leaf), rows
        return plan

    # general program. If its second derivatives enter only as a constant-weighted sum (the Laplacian part of a nonlinear equation),
    # the kernels propagate that ONE combined stream -- layout [u, firsts (nd), combined], as for affine residuals -- and the
    # program reads it as a register of its own
    comb_w = None
    if spec.n2 >= 1 and spec.nd >= 2 and spec.n3 == 0 and spec.nd <= MAX_DIRS:
        split = _second_order_split(root, spec, {})
        if split is not None:
            weights = [float(split[0].get(1 + spec.nd + k, 0.0)) if k < spec.n2 else 0.0 for k in range(spec.nd)]
            if _worth_combining(weights, spec):
                comb_w = weights
                root = _s_add(Sym('stream', alpha=('comb',)), split[1])
                S = spec.nd + 2
    # maximal x-only sub-expressions (with at least one op) become pre-pass rows
    uses = {}
    aux_of = {}

    def collect(node):
        if not _uses_streams(node, uses):
            if node.kind == 'op' and id(node) not in aux_of:
                aux_of[id(node)] = aux_row(node)
            return
        for a in node.args:
            collect(a)
    collect(root)
    ic_row = ic_const = None
    if ic_root is not None:
        # the callable IC joins the pre-pass BEFORE the program numbers its registers (they start behind the rows)
        saved = (list(pre.code), list(pre.consts), dict(pre.memo), dict(pre._cidx), list(rows))
        try:
            ic_row, ic_const = _emit_ic_rows(pre, pre_leaf, rows, spec, ic_root, comb_w)
        except TraceUnsupported:
            pre.code[:], pre.consts[:] = saved[0], saved[1]
            pre.memo.clear(); pre.memo.update(saved[2])
            pre._cidx.clear(); pre._cidx.update(saved[3])
            rows[:] = saved[4]
            ic_row = ic_const = None
    n_aux = len(rows)
    slots = set()

    def find_vars(node, seen):
        if id(node) in seen:
            return
        seen.add(id(node))
        if node.kind == 'var':
            slots.add(node.col)
        for a in node.args:
            find_vars(a, seen)
    find_vars(root, set())
    n_vars = max(slots) + 1 if slots else 0
    # (the library runs the step on the smallest COMPILED stream shape that holds this one and re-bases every register behind the streams
    #  by the difference -- pinn_abi.cpp pick_n2 / "re-basing": the budget of a program is what is left after that shift. Round 6: a
    #  random-equation soak met a 38-register program on an (nd, n2) = (2, 1) spec, which runs on the (2, 2) kernels: refused by the library
    #  inside the first fit call instead of here, where a refusal means the generic path)
    main = _Emitter(first_temp=S + n_inputs + n_aux + n_vars, limit=MAX_REGS - kernel_stream_shift(spec.nd, spec.n2p))

    def main_leaf(node):
        if id(node) in aux_of:
            return S + n_inputs + aux_of[id(node)]
        if node.kind == 'var':
            return S + n_inputs + n_aux + node.col
        if node.kind == 'stream':
            if node.alpha == ('comb',):
                return 1 + spec.nd
            if node.alpha not in spec.index:
                raise TraceUnsupported(f'stream {node.alpha} not in {spec}')
            if comb_w is not None and spec.index[node.alpha] > spec.nd:
                raise TraceUnsupported('internal: second-order stream beside the combined one')
            return spec.index[node.alpha]
        return S + node.col if node.kind == 'input' else None

    res = main.visit(root, main_leaf)
    if not main.code or main.code[-1][1] != res:
        main.emit('COPY', res)                      # the residual must be the value of the last instruction
    plan = ResidualPlan(RES_PROGRAM, n_inputs, S, pre.finish(), n_aux, program=(main.code, main.consts), n_vars=n_vars)
    plan._pre_emitter, plan._rows = (pre, pre_leaf), rows
    plan.ic_row, plan.ic_const = ic_row, ic_const
    plan.comb_w = comb_w
    return plan


def compile_program(root, spec, n_inputs):
    """ plain single program over streams and inputs (no pre-pass); kept for tests of the register convention. """
    S = spec.n_streams
    em = _Emitter(first_temp=S + n_inputs)

    def leaf(node):
        if node.kind == 'stream':
            if node.alpha not in spec.index:
                raise TraceUnsupported(f'stream {node.alpha} not in {spec}')
            return spec.index[node.alpha]
        return S + node.col if node.kind == 'input' else None

    res = em.visit(root, leaf)
    if not em.code or em.code[-1][1] != res:
        em.emit('COPY', res)
    return em.code, em.consts


def _run_code_numpy(code, consts, regs, n, aux=None):
    inv = {v: k for k, v in OPS.items()}
    out = None
    for op, dst, a, b in code:
        name = inv[op]
        if name == 'STORE':
            aux[b] = regs[a]
            continue
        if name == 'CONST':
            out = np.full(n, consts[a], dtype=np.float64)
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


def run_residual_numpy(plan, streams, xs, var_values=None):
    """ fp64 host interpreter of a ResidualPlan (validation of the trace; never on the step path). `var_values`: current
    values of the user slots 0..n_vars-1. """
    n, d, S = xs.shape[0], xs.shape[1], streams.shape[0]
    aux = {}
    if plan.n_aux:
        regs = {c: xs[:, c].astype(np.float64) for c in range(d)}
        _run_code_numpy(plan.pre[0], plan.pre[1], regs, n, aux)
    if plan.comb_w is not None:
        # streams arrive in the caller's layout [u, firsts, seconds...]: fold the seconds into the combined stream
        nd = len(plan.coef) - 2
        comb = sum(plan.comb_w[k] * streams[1 + nd + k] for k in range(S - 1 - nd))
        streams = np.concatenate([streams[:1 + nd], comb[None, :]], axis=0)
        S = streams.shape[0]
    if plan.kind == RES_AFFINE:
        r = aux[plan.src_row].copy() if plan.src_row >= 0 else np.full(n, plan.src_const, dtype=np.float64)
        for s in range(S):
            c = aux[plan.coef_row[s]] if plan.coef_row[s] >= 0 else plan.coef[s]
            r = r + c * streams[s]
        return r
    regs = {s: streams[s].astype(np.float64) for s in range(S)}
    for c in range(d):
        regs[S + c] = xs[:, c].astype(np.float64)
    for m in range(plan.n_aux):
        regs[S + d + m] = aux[m]
    for k in range(plan.n_vars):
        regs[S + d + plan.n_aux + k] = np.full(n, float(var_values[k]), dtype=np.float64)
    return _run_code_numpy(plan.program[0], plan.program[1], regs, n)


def run_program_numpy(code, consts, streams, xs):
    """ fp64 host interpreter of a plain program from `compile_program`. """
    S, d = streams.shape[0], xs.shape[1]
    regs = {s: streams[s].astype(np.float64) for s in range(S)}
    for c in range(d):
        regs[S + c] = xs[:, c].astype(np.float64)
    return _run_code_numpy(code, consts, regs, xs.shape[0])
