""" ctypes binding of the C-ABI in include/pinn.h (libpinn_hip.so, hand-written HIP for gfx950).

This is the only place where Python meets the kernels. There is NO fallback: if the HIP library is missing the
import of the product fails loudly (`load_library`). Buffers are torch tensors; the library receives raw device
pointers (`Tensor.data_ptr()`) and torch's current HIP stream, allocates nothing and never synchronises.
"""
import contextlib
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = 'libpinn_hip.so'

MAX_LAYERS, MAX_INPUTS, MAX_DIRS, EXTRA_SLOTS = 32, 8, 4, 16
MAX_OPS, MAX_CONSTS, MAX_REGS = 64, 32, 40
MAX_STREAMS, MAX_AUX, MAX_VARS = 7, 8, 8
SAMPLE_UNIFORM, SAMPLE_NORMAL, SAMPLE_CONST = 0, 1, 2
RES_PROGRAM, RES_AFFINE = 0, 1
SKIP_PRE = 0x100         # include/pinn.h PINN_SKIP_PRE
ACT_CODES = {'tanh': 0, 'sigmoid': 1, 'sin': 2, 'identity': 3, 'softplus': 4, 'silu': 5, 'swish': 5, 'gelu': 6,
             # round 5 (include/pinn.h PINN_ACT_RELU ..): torch-default forms
             'relu': 7, 'leakyrelu': 8, 'elu': 9, 'softsign': 10, 'gelu_tanh': 11, 'mish': 12, 'selu': 13, 'tanhshrink': 14,
             'logsigmoid': 15}

OPS = dict(CONST=0, ADD=1, SUB=2, MUL=3, DIV=4, NEG=5, SIN=6, COS=7, EXP=8, LOG=9, TANH=10, SQRT=11, POW=12,
           ABS=13, SIGMOID=14, RECIP=15, COPY=16, STORE=17)


class Layout(ctypes.Structure):
    _fields_ = [(name, ctypes.c_int) for name in
                ('hp', 'lh', 'd', 'off_w1', 'off_b1', 'off_wh', 'hidden_stride', 'off_wl', 'off_bl',
                 'off_log_scale', 'off_loss', 'p_core', 'off_extra', 'p_total')]


class Program(ctypes.Structure):
    _fields_ = [('n_ops', ctypes.c_int), ('n_consts', ctypes.c_int),
                ('code', ctypes.c_uint32 * MAX_OPS), ('consts', ctypes.c_float * MAX_CONSTS)]

    @classmethod
    def from_lists(cls, code, consts):
        if len(code) > MAX_OPS or len(consts) > MAX_CONSTS:
            raise ValueError(f'residual program too long ({len(code)} ops, {len(consts)} constants)')
        prog = cls()
        prog.n_ops, prog.n_consts = len(code), len(consts)
        for i, (op, dst, a, b) in enumerate(code):
            prog.code[i] = op | (dst << 8) | (a << 16) | (b << 24)
        for i, c in enumerate(consts):
            prog.consts[i] = c
        return prog


class Residual(ctypes.Structure):
    """ pinn_residual_t """
    _fields_ = [('kind', ctypes.c_int), ('n_aux', ctypes.c_int), ('pre', Program), ('program', Program),
                ('coef', ctypes.c_float * MAX_STREAMS), ('coef_row', ctypes.c_int * MAX_STREAMS),
                ('src_const', ctypes.c_float), ('src_row', ctypes.c_int),
                ('combined', ctypes.c_int), ('comb_w', ctypes.c_float * MAX_DIRS), ('n_vars', ctypes.c_int),
                ('ic_var1', ctypes.c_int), ('ic_rows', ctypes.c_int), ('ic_row', ctypes.c_int * MAX_STREAMS),
                ('ic_cst', ctypes.c_float * MAX_STREAMS), ('pre_consts64', ctypes.c_double * MAX_CONSTS)]

    @classmethod
    def build(cls, kind, n_aux, pre, program=None, coef=(), coef_row=(), src_const=0.0, src_row=-1, comb_w=None, n_vars=0,
              ic_row=None, ic_const=None):
        res = cls()
        res.kind, res.n_aux, res.n_vars = kind, n_aux, n_vars
        res.pre = Program.from_lists(*pre) if pre is not None else Program()
        if pre is not None:                         # the pre-pass runs in fp64: its constants unrounded (include/pinn.h)
            for i, c in enumerate(pre[1]):
                res.pre_consts64[i] = float(c)
        res.program = Program.from_lists(*program) if program is not None else Program()
        for i in range(MAX_STREAMS):
            res.coef[i] = coef[i] if i < len(coef) else 0.0
            res.coef_row[i] = coef_row[i] if i < len(coef_row) else -1
        res.src_const, res.src_row = src_const, src_row
        res.combined = 1 if comb_w is not None else 0
        for i in range(MAX_DIRS):
            res.comb_w[i] = comb_w[i] if comb_w is not None and i < len(comb_w) else 0.0
        res.ic_rows = 0 if ic_row is None else 1
        for i in range(MAX_STREAMS):
            res.ic_row[i] = ic_row[i] if ic_row is not None and i < len(ic_row) else -1
            res.ic_cst[i] = ic_const[i] if ic_const is not None and i < len(ic_const) else 0.0
        return res


def bind(lib):
    """ declare the signatures of include/pinn.h on a loaded shared library. """
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    ip = ctypes.POINTER(ctypes.c_int)
    lib.pinn_last_error.restype = ctypes.c_char_p
    lib.pinn_backend.restype = ctypes.c_char_p
    lib.pinn_create.argtypes = [ip, i32, i32, i32, i32, i32, i32, ctypes.POINTER(f32), ctypes.POINTER(f32), f32,
                                ctypes.POINTER(vp)]
    lib.pinn_create_ex.argtypes = [ip, i32, ip, i32, ip, ip, i32, i32, i32, i32, ctypes.POINTER(f32), ctypes.POINTER(f32),
                                   f32, ctypes.POINTER(vp)]
    lib.pinn_destroy.argtypes = [vp]
    lib.pinn_set_act_params.argtypes = [vp, ctypes.POINTER(f32), i32]
    lib.pinn_layout.argtypes = [vp, ctypes.POINTER(Layout)]
    lib.pinn_workspace_bytes.argtypes = [vp, i64, i32, i32]
    lib.pinn_workspace_bytes.restype = ctypes.c_size_t
    lib.pinn_jet_forward.argtypes = [vp, vp, vp, i64, ip, i32, i32, vp, f32, vp, vp]
    lib.pinn_jet_forward_ws.argtypes = [vp, vp, vp, i64, ip, i32, i32, vp, f32, vp, vp, ctypes.c_size_t, vp]
    lib.pinn_jet_backward.argtypes = [vp, vp, vp, i64, ip, i32, i32, vp, f32, vp, vp, i32, vp, ctypes.c_size_t, vp]
    lib.pinn_residual_step.argtypes = [vp, ctypes.POINTER(Residual), vp, vp, i64, ip, i32, i32, vp, f32, f32, vp, vp,
                                       ctypes.c_size_t, vp]
    lib.pinn_residual_step_add.argtypes = lib.pinn_residual_step.argtypes
    lib.pinn_adam_step.argtypes = [vp, vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, vp]
    lib.pinn_adam_step_at.argtypes = [vp, vp, vp, vp, vp, i64, vp, i32, f32, f32, f32, f32, vp, i32, vp]
    lib.pinn_residual_adam_step.argtypes = [vp, ctypes.POINTER(Residual), vp, vp, i64, ip, i32, i32, vp, f32, vp, vp, vp, vp,
                                            vp, i32, f32, f32, f32, f32, vp, vp, ctypes.c_size_t, vp]
    lib.pinn_sample_points.argtypes = [vp, i64, i32, ip, ctypes.POINTER(f32), ctypes.POINTER(f32), ctypes.c_uint64,
                                       ctypes.c_uint64, vp]
    lib.pinn_fit_steps.argtypes = [vp, ctypes.POINTER(Residual), vp, vp, i64, ip, ctypes.POINTER(f32), ctypes.POINTER(f32),
                                   ctypes.c_uint64, ctypes.c_uint64, ip, i32, i32, f32, vp, vp, vp, vp, vp, i32, f32, f32, f32, f32,
                                   vp, i32, vp, ctypes.c_size_t, vp]
    lib.pinn_fit_steps.restype = i32
    lib.pinn_fit_steps_graph.argtypes = [vp, ctypes.POINTER(Residual), vp, vp, i64, ip, ctypes.POINTER(f32), ctypes.POINTER(f32),
                                         ctypes.c_uint64, ctypes.c_uint64, ip, i32, i32, f32, vp, vp, vp, vp, vp, i32, f32, f32, f32, f32,
                                         vp, i32, vp, ctypes.c_size_t, vp, ctypes.c_size_t, vp]
    lib.pinn_fit_steps_graph.restype = i32
    lib.pinn_fit_ctrl_bytes.restype = ctypes.c_size_t
    lib.pinn_set_tanh_mode.argtypes = [vp, i32]
    lib.pinn_set_tanh_mode.restype = i32
    lib.pinn_set_gemm_mode.argtypes = [vp, i32]
    lib.pinn_set_gemm_mode.restype = i32
    lib.pinn_profile_tile.argtypes = [i32]
    lib.pinn_profile_tile.restype = i32
    lib.pinn_last_tile_ms.restype = f32
    lib.pinn_last_wgrad_ms.restype = f32
    lib.pinn_last_kernel_name.restype = ctypes.c_char_p
    lib.pinn_last_wgrad_kernel_name.restype = ctypes.c_char_p
    if hasattr(lib, 'pinn_debug_phase_buffer'):                  # -DPINN_DEBUG_ABI experiment builds only
        lib.pinn_debug_phase_buffer.argtypes = [vp]
    lib.pinn_debug_wgx_chunk_bytes.argtypes = [vp, ctypes.c_longlong]
    lib.pinn_debug_max_wgs_per_cu.argtypes = [vp, ctypes.c_int]
    lib.pinn_debug_prepass_in_kernel.argtypes = [vp, ctypes.c_int]
    lib.pinn_debug_fit_persistent.argtypes = [vp, ctypes.c_int]
    lib.pinn_fit_chunk_status.argtypes = []
    lib.pinn_fit_chunk_status.restype = ctypes.c_int
    lib.pinn_debug_fit_onecu_rounds.argtypes = [vp, ctypes.c_int]       # (a required symbol like the others: ABI_SYMBOLS)
    lib.pinn_last_launch_info.argtypes = [ctypes.POINTER(ctypes.c_int32)]
    lib.pinn_debug_fit_graph_stats.argtypes = [ctypes.POINTER(ctypes.c_int32)]
    for name in ('pinn_create', 'pinn_create_ex', 'pinn_destroy', 'pinn_layout', 'pinn_jet_forward', 'pinn_jet_forward_ws', 'pinn_jet_backward',
                 'pinn_residual_step', 'pinn_residual_adam_step', 'pinn_adam_step', 'pinn_adam_step_at'):
        getattr(lib, name).restype = i32
    return lib


ABI_SYMBOLS = ('pinn_create', 'pinn_create_ex', 'pinn_destroy', 'pinn_layout', 'pinn_workspace_bytes', 'pinn_jet_forward', 'pinn_jet_forward_ws',
               'pinn_jet_backward', 'pinn_residual_step', 'pinn_residual_step_add', 'pinn_residual_adam_step', 'pinn_adam_step', 'pinn_adam_step_at', 'pinn_sample_points', 'pinn_fit_steps', 'pinn_fit_steps_graph', 'pinn_fit_ctrl_bytes', 'pinn_set_gemm_mode', 'pinn_set_tanh_mode', 'pinn_profile_tile',
               'pinn_last_tile_ms', 'pinn_last_wgrad_ms', 'pinn_last_kernel_name', 'pinn_last_wgrad_kernel_name', 'pinn_debug_last_kernel',
               'pinn_debug_prepass_in_kernel', 'pinn_debug_wgx_chunk_bytes', 'pinn_debug_max_wgs_per_cu', 'pinn_debug_fit_persistent', 'pinn_fit_chunk_status', 'pinn_set_act_params', 'pinn_debug_fit_onecu_rounds', 'pinn_debug_fit_graph_stats', 'pinn_last_launch_info',
               'pinn_last_error', 'pinn_backend')

_LIB = None


def library_path():
    return os.path.join(_HERE, LIB_NAME)


def load_library():
    """ the product library; raises if it has not been built (no CPU / eager fallback exists). """
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(f'{path} is missing: build it with `python -m pydens_amd.csrc.build` '
                               '(or __graft_entry__.build()); pydens_amd has no fallback path')
        lib = bind(ctypes.CDLL(path))
        if lib.pinn_backend() != b'hip-gfx950':
            raise RuntimeError(f'{path} is not the HIP build ({lib.pinn_backend()!r})')
        _LIB = lib
    return _LIB


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t):
    if t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None


stream_of = _stream

_NO_GUARD = contextlib.nullcontext()


def _on_device(t):
    """ the library launches on the CURRENT HIP device (grid sizing, kernel attributes, the stream's owner): make the
    tensor's device current for the call when it is not (`Solver(device='cuda:1')` while cuda:0 is current) """
    if t.is_cuda and torch.cuda.current_device() != t.device.index:
        return torch.cuda.device(t.device)
    return _NO_GUARD


def _check(t, name, dtype=torch.float32):
    if t is None:
        return
    if t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f'{name} must be a contiguous {dtype} tensor')


class Net:
    """ pinn_t handle + its flat padded parameter layout. """
    def __init__(self, layer_dims, activation, ndims, nparams=0, has_bc=False, bc_value=0.0, has_ic=False,
                 domain=None, lib=None, skips=()):
        """ activation: one name for every hidden layer or a sequence with one name per hidden layer;
        skips: (src, dst[, pre[, src_pre]]) hidden-layer indices: output of dst += output of src ('R ... +' layouts); pre: the
        sum enters in front of the activation of dst ('faR fa f+ a'); src_pre: the pre-activation of src is what travels ('fRa'). """
        self.lib = lib if lib is not None else load_library()
        n_hidden = len(layer_dims) - 2
        names = [activation] * n_hidden if isinstance(activation, str) else list(activation)
        if len(names) != n_hidden:
            raise ValueError(f'{n_hidden} hidden layers need {n_hidden} activations, got {names}')
        # 'Name:value': an activation with its one parameter (LeakyReLU negative_slope / ELU alpha / Softplus beta; include/pinn.h
        # pinn_set_act_params) -- what model._activation_name makes of a module instance configured away from torch's default
        par_default = {'leakyrelu': 0.01, 'elu': 1.0, 'softplus': 1.0}
        pars = []
        for i, name in enumerate(names):
            base, _, value = str(name).partition(':')
            if value and base.lower() not in par_default:
                raise NotImplementedError(f'activation {name!r}: only LeakyReLU, ELU and Softplus take a parameter')
            pars.append(float(value) if value else par_default.get(base.lower(), 0.0))
            names[i] = base
        codes = [ACT_CODES.get(str(name).lower()) for name in names]
        if None in codes:
            raise NotImplementedError(f'activation {names[codes.index(None)]!r}: the HIP kernels implement '
                                      + ', '.join(sorted(ACT_CODES)) + ' (torch-default forms)')
        skips = sorted((s[0], s[1], bool(s[2]) if len(s) > 2 else False, bool(s[3]) if len(s) > 3 else False) for s in skips)
        # nested skips (another 'R' while one is open): their forward passes need scratch (include/pinn.h pinn_jet_forward_ws)
        self.nested = any(a[0] < b[0] < a[1] for a in skips for b in skips if a is not b)
        self._fwd_ws = None
        # second set of full breadth kernels (pinn_inst.inc pinn_needs_allact): activation codes above 7 or nested skips
        self.allact = self.nested or any(c > 7 for c in codes)
        domain = list(domain) if domain is not None else [(0.0, 1.0)] * ndims
        dims = (ctypes.c_int * len(layer_dims))(*layer_dims)
        lo = (ctypes.c_float * ndims)(*[float(d[0]) for d in domain])
        hi = (ctypes.c_float * ndims)(*[float(d[1]) for d in domain])
        handle = ctypes.c_void_p()
        acts = (ctypes.c_int * max(1, n_hidden))(*codes)
        src = (ctypes.c_int * max(1, len(skips)))(*[s | (SKIP_PRE if src_pre else 0) for s, _, _, src_pre in skips])
        dst = (ctypes.c_int * max(1, len(skips)))(*[d | (SKIP_PRE if pre else 0) for _, d, pre, _ in skips])
        rc = self.lib.pinn_create_ex(dims, len(layer_dims) - 1, acts, len(skips), src, dst, ndims, nparams, int(has_bc),
                                     int(has_ic), lo, hi, float(bc_value), ctypes.byref(handle))
        self._raise(rc)
        self.handle = handle
        self.act_params = pars
        if any(p != par_default.get(str(n).lower(), 0.0) for p, n in zip(pars, names)):
            self._raise(self.lib.pinn_set_act_params(self.handle, (ctypes.c_float * max(1, n_hidden))(*pars), n_hidden))
        lay = Layout()
        self._raise(self.lib.pinn_layout(self.handle, ctypes.byref(lay)))
        self.layout = lay
        self.layer_dims = list(layer_dims)
        self.ndims, self.nparams = ndims, nparams
        self.gemm_mode = 'fp32'
        self.tanh_mode = 'fast'
        if os.environ.get('PYDENS_AMD_GEMM'):
            self.set_gemm_mode(os.environ['PYDENS_AMD_GEMM'])
        # small fit chunks of narrow nets as ONE launch each (include/pinn.h pinn_debug_fit_persistent): '2' one-CU form for batches of a few
        # tiles (the default), '1' grid form (measured slower than launch graphs on MI355X), '0' never
        if os.environ.get('PYDENS_AMD_FIT_PERSIST') in ('0', '1', '2'):
            self.lib.pinn_debug_fit_persistent(self.handle, int(os.environ['PYDENS_AMD_FIT_PERSIST']))
        if os.environ.get('PYDENS_AMD_FIT_ROUNDS'):
            try:
                rounds = int(os.environ['PYDENS_AMD_FIT_ROUNDS'])
            except ValueError:
                raise ValueError(f"PYDENS_AMD_FIT_ROUNDS={os.environ['PYDENS_AMD_FIT_ROUNDS']!r}: expected an integer") from None
            self.lib.pinn_debug_fit_onecu_rounds(self.handle, rounds)
        if os.environ.get('PYDENS_AMD_TANH'):
            self.set_tanh_mode(os.environ['PYDENS_AMD_TANH'])
        if os.environ.get('PYDENS_AMD_WGX_CHUNK_MB'):           # experiments: slab budget of the widths >= 128 (pinn_debug_wgx_chunk_bytes)
            self.lib.pinn_debug_wgx_chunk_bytes(self.handle, int(float(os.environ['PYDENS_AMD_WGX_CHUNK_MB']) * (1 << 20)))

    def set_tanh_mode(self, mode):
        """ 'fast' (default) or 'accurate' (polynomial tanh below |z| = 0.45 where the kernel has that form: include/pinn.h pinn_set_tanh_mode) """
        codes = {'fast': 0, 'accurate': 1}
        if mode not in codes:
            raise ValueError(f'tanh mode {mode!r}: expected one of {sorted(codes)}')
        self._raise(self.lib.pinn_set_tanh_mode(self.handle, codes[mode]))
        self.tanh_mode = mode

    def set_gemm_mode(self, mode):
        """ 'fp32' (default: exact-fp32 MFMA) or 'bf16x3' (fp32 operands as three bf16, six products, fp32 accumulate: the
        split-bf16 kernels where they exist, include/pinn.h pinn_set_gemm_mode) """
        codes = {'fp32': 0, 'bf16x3': 1}
        if mode not in codes:
            raise ValueError(f'gemm mode {mode!r}: expected one of {sorted(codes)}')
        self._raise(self.lib.pinn_set_gemm_mode(self.handle, codes[mode]))
        self.gemm_mode = mode

    def _raise(self, rc):
        if rc != 0:
            raise RuntimeError('libpinn: ' + self.lib.pinn_last_error().decode())

    def __del__(self):
        if getattr(self, 'handle', None):
            self.lib.pinn_destroy(self.handle)
            self.handle = None

    # ---- parameter views ----------------------------------------------------------------------------------------
    def param_views(self, flat):
        """ nn.Linear-shaped (weight [out,in], bias [out]) views into the flat padded buffer, layer by layer. """
        L, dims = self.layout, self.layer_dims
        n_lin = len(dims) - 1
        views = []
        for l in range(n_lin):
            n_in, n_out = dims[l], dims[l + 1]
            if l == 0:
                w = flat.as_strided((n_out, n_in), (L.d, 1), L.off_w1)
                b = flat.as_strided((n_out,), (1,), L.off_b1)
            elif l == n_lin - 1:
                w = flat.as_strided((1, n_in), (L.hp, 1), L.off_wl)
                b = flat.as_strided((1,), (1,), L.off_bl)
            else:
                off = L.off_wh + (l - 1) * L.hidden_stride
                w = flat.as_strided((n_out, n_in), (L.hp, 1), off)
                b = flat.as_strided((n_out,), (1,), off + L.hp * L.hp)
            views.append((w, b))
        return views

    # ---- kernels ----------------------------------------------------------------------------------------------------
    def _dirs(self, dir_cols):
        dir_cols = list(dir_cols)
        return (ctypes.c_int * max(len(dir_cols), 1))(*dir_cols), len(dir_cols)

    def workspace_bytes(self, n_points, nd, n2):
        return int(self.lib.pinn_workspace_bytes(self.handle, n_points, nd, n2))

    def jet_forward(self, params, xs, dir_cols=(), n2=0, ic_streams=None, ic_const=0.0, out=None):
        _check(params, 'params'); _check(xs, 'xs'); _check(ic_streams, 'ic_streams')
        n = xs.shape[0]
        dirs, nd = self._dirs(dir_cols)
        s = 1 + nd + (n2 & 7) + ((n2 >> 3) & 7) + (n2 >> 6)       # n2 may be packed: seconds | thirds << 3 | fourths << 6 (include/pinn.h)
        if out is None:
            out = torch.empty((s, n), dtype=torch.float32, device=xs.device)
        _check(out, 'out')
        with _on_device(params):
            if self.nested:
                need = self.workspace_bytes(n, nd, n2)
                if self._fwd_ws is None or self._fwd_ws.numel() * 4 < need or self._fwd_ws.device != xs.device:
                    self._fwd_ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=xs.device)
                self._raise(self.lib.pinn_jet_forward_ws(self.handle, _ptr(params), _ptr(xs), n, dirs, nd, n2, _ptr(ic_streams),
                                                         float(ic_const), _ptr(out), _ptr(self._fwd_ws), self._fwd_ws.numel() * 4,
                                                         _stream(xs)))
            else:
                self._raise(self.lib.pinn_jet_forward(self.handle, _ptr(params), _ptr(xs), n, dirs, nd, n2, _ptr(ic_streams),
                                                      float(ic_const), _ptr(out), _stream(xs)))
        return out

    def jet_backward(self, params, xs, grad_streams, grads, workspace, dir_cols=(), n2=0, ic_streams=None,
                     ic_const=0.0, accumulate=False):
        for t, name in ((params, 'params'), (xs, 'xs'), (grad_streams, 'grad_streams'), (grads, 'grads'),
                        (ic_streams, 'ic_streams')):
            _check(t, name)
        dirs, nd = self._dirs(dir_cols)
        with _on_device(params):
            self._raise(self.lib.pinn_jet_backward(self.handle, _ptr(params), _ptr(xs), xs.shape[0], dirs, nd, n2,
                                                   _ptr(ic_streams), float(ic_const), _ptr(grad_streams), _ptr(grads),
                                                   int(accumulate), _ptr(workspace), workspace.numel() * workspace.element_size(),
                                                   _stream(xs)))

    def residual_step(self, residual, params, xs, grads, workspace, dir_cols=(), n2=0, ic_streams=None, ic_const=0.0,
                      inv_n_global=None, accumulate=False, stream=None):
        for t, name in ((params, 'params'), (xs, 'xs'), (grads, 'grads'), (ic_streams, 'ic_streams')):
            _check(t, name)
        dirs, nd = self._dirs(dir_cols)
        n = xs.shape[0]
        inv_n = 1.0 / n if inv_n_global is None else inv_n_global
        fn = self.lib.pinn_residual_step_add if accumulate else self.lib.pinn_residual_step
        with _on_device(params):
            self._raise(fn(self.handle, ctypes.byref(residual), _ptr(params), _ptr(xs), n, dirs, nd,
                                                    n2, _ptr(ic_streams), float(ic_const), float(inv_n), _ptr(grads),
                                                    _ptr(workspace), workspace.numel() * workspace.element_size(),
                                                    _stream(xs) if stream is None else stream))

    def residual_adam_step(self, residual, params, xs, grads, workspace, exp_avg, exp_avg_sq, mask, step_tensor, step,
                           lr, betas=(0.9, 0.999), eps=1e-8, dir_cols=(), n2=0, ic_streams=None, ic_const=0.0,
                           loss_out=None, stream=None):
        """ `loss_out`: device ADDRESS (int) that also receives the loss of the step; `stream`: handle from `stream_of`
        when the caller has looked it up already (a fit loop asks once, not per iteration). """
        for t, name in ((params, 'params'), (xs, 'xs'), (grads, 'grads'), (ic_streams, 'ic_streams'),
                        (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq')):
            _check(t, name)
        _check(mask, 'mask', torch.uint8)
        _check(step_tensor, 'step', torch.int32)
        dirs, nd = self._dirs(dir_cols)
        with _on_device(params):
            self._raise(self.lib.pinn_residual_adam_step(
                self.handle, ctypes.byref(residual), _ptr(params), _ptr(xs), xs.shape[0], dirs, nd, n2, _ptr(ic_streams),
                float(ic_const), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(mask), _ptr(step_tensor), int(step),
                float(lr), float(betas[0]), float(betas[1]), float(eps),
                None if loss_out is None else ctypes.c_void_p(loss_out), _ptr(workspace),
                workspace.numel() * workspace.element_size(), _stream(xs) if stream is None else stream))

    def fit_steps(self, residual, params, xs, columns, seed, call_index0, grads, workspace, exp_avg, exp_avg_sq, mask,
                  step_tensor, step0, lr, betas, eps, loss_history, k_steps, dir_cols=(), n2=0, ic_const=0.0, stream=None, ctrl=None):
        """ `k_steps` iterations of the fit loop (sample -> fused step -> Adam) enqueued by one call (include/pinn.h
        pinn_fit_steps); `xs` is the [N, d] batch buffer every iteration overwrites, `loss_history` a float32 device tensor
        with at least k_steps entries. """
        for t, name in ((params, 'params'), (xs, 'xs'), (grads, 'grads'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq'),
                        (loss_history, 'loss_history')):
            _check(t, name)
        _check(mask, 'mask', torch.uint8)
        _check(step_tensor, 'step', torch.int32)
        d = len(columns)
        if xs.dim() != 2 or xs.shape[1] != d or loss_history.numel() < k_steps:
            raise ValueError(f'xs must be [N, {d}] and loss_history hold {k_steps} entries')
        kind = (ctypes.c_int * d)(*[int(c[0]) for c in columns])
        a = (ctypes.c_float * d)(*[float(c[1]) for c in columns])
        b = (ctypes.c_float * d)(*[float(c[2]) for c in columns])
        dirs, nd = self._dirs(dir_cols)
        common = (self.handle, ctypes.byref(residual), _ptr(params), _ptr(xs), xs.shape[0], kind, a, b,
                  int(seed) & (2 ** 64 - 1), int(call_index0), dirs, nd, n2, float(ic_const), _ptr(grads), _ptr(exp_avg),
                  _ptr(exp_avg_sq), _ptr(mask), _ptr(step_tensor), int(step0), float(lr), float(betas[0]), float(betas[1]),
                  float(eps), _ptr(loss_history), int(k_steps), _ptr(workspace), workspace.numel() * workspace.element_size())
        with _on_device(params):
            if ctrl is not None:
                # `ctrl`: a uint8 device tensor of pinn_fit_ctrl_bytes() bytes -- the chunk as one replayable launch graph
                # (pinn_fit_steps_graph; the library falls back to the eager loop by itself where a graph does not apply)
                self._raise(self.lib.pinn_fit_steps_graph(*common, _ptr(ctrl), ctrl.numel() * ctrl.element_size(),
                                                          _stream(xs) if stream is None else stream))
            else:
                self._raise(self.lib.pinn_fit_steps(*common, _stream(xs) if stream is None else stream))

    def sample_points(self, xs, columns, seed, call_index, stream=None):
        """ fill xs [N, d] on the device: columns = [(kind, a, b), ...] with kind SAMPLE_UNIFORM (a + (b - a) u),
        SAMPLE_NORMAL (a + b z) or SAMPLE_CONST (a); Philox4x32-10 keyed by (seed, call_index), include/pinn.h. """
        _check(xs, 'xs')
        d = len(columns)
        if xs.dim() != 2 or xs.shape[1] != d:
            raise ValueError(f'xs must be [N, {d}]')
        kind = (ctypes.c_int * d)(*[int(c[0]) for c in columns])
        a = (ctypes.c_float * d)(*[float(c[1]) for c in columns])
        b = (ctypes.c_float * d)(*[float(c[2]) for c in columns])
        with _on_device(xs):
            self._raise(self.lib.pinn_sample_points(_ptr(xs), xs.shape[0], d, kind, a, b, int(seed) & (2 ** 64 - 1),
                                                    int(call_index), _stream(xs) if stream is None else stream))
        return xs

    def adam_step(self, params, grads, exp_avg, exp_avg_sq, mask, step, lr, betas=(0.9, 0.999), eps=1e-8, at=0,
                  loss_out=None, stream=None):
        """ `step`: int32 device counter; at > 0: the host's 1-based step count (one launch, counter mirrored);
        loss_out: device ADDRESS (int) that receives grads[off_loss] in the same launch (at > 0 only) """
        for t, name in ((params, 'params'), (grads, 'grads'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq')):
            _check(t, name)
        _check(mask, 'mask', torch.uint8)
        _check(step, 'step', torch.int32)
        if at > 0:
            with _on_device(params):
                self._raise(self.lib.pinn_adam_step_at(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(mask),
                                                       params.numel(), _ptr(step), int(at), float(lr), float(betas[0]),
                                                       float(betas[1]), float(eps),
                                                       None if loss_out is None else ctypes.c_void_p(loss_out),
                                                       int(self.layout.off_loss), _stream(params) if stream is None else stream))
            return
        with _on_device(params):
            self._raise(self.lib.pinn_adam_step(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(mask),
                                                params.numel(), _ptr(step), float(lr), float(betas[0]), float(betas[1]),
                                                float(eps), _stream(params)))
