""" TEST INFRASTRUCTURE ONLY -- numpy fp64 restatement of the *kernel-side* mathematics.

The reference obtains du/dx_i and d2u/dx_i2 by nested reverse-mode autograd (model_torch.py:174-178) and the
parameter gradient by one more reverse sweep (`loss.backward()`, :460). The HIP engine computes the same
quantities by forward Taylor-mode "jets" through the MLP plus one hand-written reverse sweep (DESIGN.md
section 3). This file is the plain-numpy, double-precision statement of exactly those formulas; the tests use it
(a) to prove the formulas equal the reference's nested autograd (vs `oracle.pinn_oracle` in fp64) and
(b) as the fp64 arbiter for the fp32 kernels.

Stream order (shared with the kernels): index 0 = value; 1..ND = first derivative along direction k;
1+ND..ND+N2 = second derivative along direction k < N2 (directions that need a second derivative come first);
then N3 third derivatives along the first N3 of those (single columns only: D(D(D(f, x), x), x), KdV-type equations).
`dir_cols[k]` is the input column that direction k differentiates, or a pair (a, b) for the diagonal direction
e_a + e_b (mixed partials: u_ab = (u_vv - u_aa - u_bb) / 2).
"""
import numpy as np


def act_param(act):
    """ 'name' or 'name:value' -> (name, value or None): the one parameter a LeakyReLU (negative_slope) / ELU (alpha) / Softplus (beta)
    module may carry (round 6; pinn_kernel.h PinnAct, include/pinn.h pinn_set_act_params) """
    base, _, value = str(act).partition(':')
    return base, (float(value) if value else None)


def act_derivs(z, act, fourth=False):
    """ activation value and its first three (four) derivatives. """
    act, par = act_param(act)
    if act == 'softplus' and par is not None and par != 1.0:
        # softplus_beta(z) = softplus(beta z) / beta: derivative n is beta^(n-1) times derivative n of softplus at beta z
        out = act_derivs(par * np.asarray(z, dtype=np.float64), 'softplus', fourth=True)
        scaled = (out[0] / par, out[1], par * out[2], par ** 2 * out[3], par ** 3 * out[4])
        return scaled if fourth else scaled[:4]
    if act == 'tanh':
        t = np.tanh(z)
        d1 = 1.0 - t * t
        d2 = -2.0 * t * d1
        d3 = d1 * (6.0 * t * t - 2.0)
        d4 = d1 * t * (16.0 - 24.0 * t * t)
    elif act == 'sigmoid':
        t = 1.0 / (1.0 + np.exp(-z))
        d1 = t * (1.0 - t)
        q = 1.0 - 2.0 * t
        d2 = d1 * q
        d3 = d1 * (q * q - 2.0 * d1)
        d4 = d1 * q * (q * q - 8.0 * d1)
    elif act == 'sin':
        t, d1 = np.sin(z), np.cos(z)
        d2, d3, d4 = -t, -d1, t
    elif act == 'identity':
        t, d1 = z, np.ones_like(z)
        d2 = d3 = d4 = np.zeros_like(z)
    elif act in ('softplus', 'silu', 'swish'):
        # s = sigmoid(z), a = s(1 - s), q = 1 - 2s:  ds = a, da = a q, dq = -2a
        sg = 1.0 / (1.0 + np.exp(-z))
        a = sg * (1.0 - sg)
        q = 1.0 - 2.0 * sg
        a3 = a * (q * q - 2.0 * a)                 # second derivative of a
        a4 = a * q * (q * q - 8.0 * a)             # third derivative of a
        if act == 'softplus':                      # log(1 + e^z): derivatives s, a, a q, a3
            t = np.logaddexp(0.0, z)
            d1, d2, d3, d4 = sg, a, a * q, a3
        else:                                      # z s
            t = z * sg
            d1 = sg + z * a
            d2 = 2.0 * a + z * a * q
            d3 = 3.0 * a * q + z * a3
            d4 = 4.0 * a3 + z * a4
    elif act == 'gelu':                            # z Phi(z) (erf form, torch.nn.GELU default)
        from math import erf, pi, sqrt
        Phi = 0.5 * (1.0 + np.vectorize(erf)(z / sqrt(2.0)))
        phi = np.exp(-0.5 * z * z) / sqrt(2.0 * pi)
        z2 = z * z
        t = z * Phi
        d1 = Phi + z * phi
        d2 = phi * (2.0 - z2)
        d3 = phi * z * (z2 - 4.0)
        d4 = phi * ((7.0 - z2) * z2 - 4.0)
    elif act in ('relu', 'leakyrelu'):             # torch: slope 0 / 0.01 (the defaults) on z <= 0, derivative taken as in torch (z > 0 ? 1 : slope)
        slope = 0.0 if act == 'relu' else (0.01 if par is None else par)
        t = np.where(z > 0, z, slope * z)
        d1 = np.where(z > 0, 1.0, slope)
        d2 = d3 = d4 = np.zeros_like(z)
    elif act in ('elu', 'selu'):                   # s (z > 0 ? z : alpha (e^z - 1)); ELU: s = alpha = 1
        s_, al = (1.0507009873554805, 1.6732632423543772) if act == 'selu' else (1.0, 1.0 if par is None else par)
        e = s_ * al * np.exp(z)
        t = np.where(z > 0, s_ * z, s_ * al * np.expm1(z))
        d1 = np.where(z > 0, s_, e)
        d2 = d3 = d4 = np.where(z > 0, 0.0, e)
    elif act == 'softsign':                        # z / (1 + |z|)
        a, sg = 1.0 / (1.0 + np.abs(z)), np.sign(z)
        t, d1, d2, d3, d4 = z * a, a ** 2, -2.0 * sg * a ** 3, 6.0 * a ** 4, -24.0 * sg * a ** 5
    elif act == 'tanhshrink':                      # z - tanh z
        _, a1, a2, a3, a4 = act_derivs(z, 'tanh', fourth=True)
        t, d1, d2, d3, d4 = z - np.tanh(z), 1.0 - a1, -a2, -a3, -a4
    elif act == 'logsigmoid':                      # log sigmoid(z) = -softplus(-z)
        sg = 1.0 / (1.0 + np.exp(-z))
        a, q = sg * (1.0 - sg), 1.0 - 2.0 * sg
        t, d1, d2, d3, d4 = -np.logaddexp(0.0, -z), 1.0 - sg, -a, -a * q, -a * (q * q - 2.0 * a)
    elif act in ('gelu_tanh', 'mish'):
        # F(z) = tanh(u(z)) by Faa di Bruno to fourth order, then (z F)^(n) = z F^(n) + n F^(n-1)
        if act == 'gelu_tanh':                     # 0.5 z (1 + tanh(k (z + c z^3)))   (torch.nn.GELU(approximate='tanh'))
            k, c = 0.7978845608028654, 0.044715
            u, u1, u2, u3, u4 = k * (z + c * z ** 3), k * (1.0 + 3.0 * c * z * z), 6.0 * k * c * z, 6.0 * k * c * np.ones_like(z), 0.0
        else:                                      # z tanh(softplus(z))               (torch.nn.Mish)
            sg = 1.0 / (1.0 + np.exp(-z))
            a, q = sg * (1.0 - sg), 1.0 - 2.0 * sg
            u, u1, u2, u3, u4 = np.logaddexp(0.0, z), sg, a, a * q, a * (q * q - 2.0 * a)
        T, a1, a2, a3, a4 = act_derivs(u, 'tanh', fourth=True)
        t1 = a1 * u1
        t2 = a2 * u1 ** 2 + a1 * u2
        t3 = a3 * u1 ** 3 + 3.0 * a2 * u1 * u2 + a1 * u3
        t4 = a4 * u1 ** 4 + 6.0 * a3 * u1 ** 2 * u2 + a2 * (4.0 * u1 * u3 + 3.0 * u2 ** 2) + a1 * u4
        sc, f0 = (0.5, 0.5 * (1.0 + T)) if act == 'gelu_tanh' else (1.0, T)
        t, d1, d2, d3, d4 = z * f0, f0 + sc * z * t1, sc * (z * t2 + 2.0 * t1), sc * (z * t3 + 3.0 * t2), sc * (z * t4 + 4.0 * t3)
    else:
        raise ValueError(act)
    return (t, d1, d2, d3, d4) if fourth else (t, d1, d2, d3)


def parse_layout(layout, activation):
    """ the letters of the reference's Block layouts (model_torch.py:142-156) that matter to a fully connected net:
    -> (activation name per hidden layer, skips [(src, dst, dst_pre, src_pre)]); 'R' / '+' sit behind a dense layer, in front of its
    activation (pre) or behind it. """
    letters = layout.replace(' ', '')
    names = None if isinstance(activation, str) else [str(a) for a in activation]
    acts, skips, layer, open_skip = [], [], -1, None
    for ch in letters:
        if ch == 'f':
            if layer >= 0 and len(acts) == layer:
                acts.append('identity')
            layer += 1
        elif ch == 'a':
            acts.append((names.pop(0) if names is not None else activation).lower())
        elif ch == 'R':
            open_skip = (layer, len(acts) == layer)
        elif ch == '+':
            skips.append((open_skip[0], layer, len(acts) == layer, open_skip[1]))
            open_skip = None
        else:
            raise ValueError(ch)
    return acts, skips


class Spec:
    """ Problem descriptor: network, ansatz and requested derivative streams. """
    def __init__(self, weights, biases, act, ndims, nparams=0, has_bc=False, bc_value=0.0, has_ic=False,
                 domain=None, log_scale=0.0, dir_cols=(), n2=0, n3=0, skips=()):
        self.W = [np.asarray(w, dtype=np.float64) for w in weights]     # [out, in] per layer (nn.Linear layout)
        self.b = [np.asarray(b, dtype=np.float64) for b in biases]
        n_hidden = len(self.W) - 1
        self.acts = [act.lower()] * n_hidden if isinstance(act, str) else [str(a).lower() for a in act]
        assert len(self.acts) == n_hidden
        self.skips = list(skips)                    # (src, dst, dst_pre, src_pre) hidden-layer indices (parse_layout)
        self.ndims, self.nparams = ndims, nparams
        self.has_bc, self.bc_value, self.has_ic = has_bc, float(bc_value), has_ic
        self.nsp = ndims - 1 if has_ic else ndims
        self.domain = [tuple(map(float, d)) for d in (domain or [(0.0, 1.0)] * ndims)]
        self.log_scale = float(log_scale)
        self.dir_cols, self.nd, self.n2, self.n3 = list(dir_cols), len(dir_cols), n2, n3
        assert n3 <= n2 <= self.nd
        self.S = 1 + self.nd + self.n2 + self.n3

    def i3(self, k):
        """ stream index of the third derivative along direction k < n3 """
        return 1 + self.nd + self.n2 + k


def _cols(direction):
    """ input columns of a differentiation direction: an int (column) or a pair (a, b) = the diagonal e_a + e_b """
    return (direction,) if isinstance(direction, (int, np.integer)) else tuple(direction)


def mlp_jet_forward(sp, xs):
    """ xs [N,d] -> net streams [S,N]; cache for the reverse sweep. """
    N = xs.shape[0]
    L = len(sp.W)
    cache = []
    # layer 1: z0 = W x + b ; z_k = W[:, col_k] ; z_kk = 0
    W, b = sp.W[0], sp.b[0]
    z = np.zeros((sp.S, N, W.shape[0]))
    z[0] = xs @ W.T + b
    for k, c in enumerate(sp.dir_cols):
        z[1 + k] = sum(W[:, j] for j in _cols(c))[None, :]
    h_prev = None
    carried = {}
    for l in range(L - 1):
        if l > 0:
            W, b = sp.W[l], sp.b[l]
            z = np.einsum('snk,ok->sno', h_prev, W)
            z[0] += b
        for k_, (src, dst, dst_pre, src_pre) in enumerate(sp.skips):      # '+' in front of the activation: z joins the carried jets
            if dst == l and dst_pre:
                z = z + carried[k_]
        for k_, (src, dst, dst_pre, src_pre) in enumerate(sp.skips):      # 'R' in front of the activation: z itself is carried
            if src == l and src_pre:
                carried[k_] = z
        t, d1, d2, d3, d4 = act_derivs(z[0], sp.acts[l], fourth=True)
        h = np.zeros_like(z)
        h[0] = t
        for k in range(sp.nd):
            h[1 + k] = d1 * z[1 + k]
        for k in range(sp.n2):
            h[1 + sp.nd + k] = d2 * z[1 + k] ** 2 + d1 * z[1 + sp.nd + k]
        for k in range(sp.n3):                                         # h''' = s' z''' + 3 s'' z' z'' + s''' z'^3
            z1, z2 = z[1 + k], z[1 + sp.nd + k]
            h[sp.i3(k)] = d1 * z[sp.i3(k)] + 3.0 * d2 * z1 * z2 + d3 * z1 ** 3
        for k_, (src, dst, dst_pre, src_pre) in enumerate(sp.skips):      # '+' behind the activation
            if dst == l and not dst_pre:
                h = h + carried[k_]
        for k_, (src, dst, dst_pre, src_pre) in enumerate(sp.skips):      # 'R' behind the activation (after a '+' that ends here)
            if src == l and not src_pre:
                carried[k_] = h
        cache.append((h_prev, z, d1, d2, d3, d4))
        h_prev = h
    W, b = sp.W[L - 1], sp.b[L - 1]
    net = np.einsum('snk,ok->sno', h_prev, W)[:, :, 0]
    net[0] += b[0]
    return net, (cache, h_prev, xs)


def mlp_jet_backward(sp, gnet, fwd_cache):
    """ gnet [S,N] = dLoss/dnet streams -> (dW list, db list). """
    cache, h_last, xs = fwd_cache
    L = len(sp.W)
    dW = [None] * L
    db = [None] * L
    dW[L - 1] = np.einsum('sn,snk->k', gnet, h_last)[None, :]
    db[L - 1] = np.array([gnet[0].sum()])
    gh = gnet[:, :, None] * sp.W[L - 1][0][None, None, :]
    gskip = {}
    for l in range(L - 2, -1, -1):
        h_prev, z, d1, d2, d3, d4 = cache[l]
        for k_, (src, dst, dst_pre, src_pre) in enumerate(sp.skips):      # reverse of the forward order at this layer
            if src == l and not src_pre:
                gh = gh + gskip[k_]
        for k_, (src, dst, dst_pre, src_pre) in enumerate(sp.skips):
            if dst == l and not dst_pre:
                gskip[k_] = gh
        gz = np.zeros_like(gh)
        acc = d1 * gh[0]
        for k in range(sp.nd):
            zk = z[1 + k]
            gz[1 + k] = d1 * gh[1 + k]
            acc = acc + d2 * zk * gh[1 + k]
            if k < sp.n2:
                zkk, ghkk = z[1 + sp.nd + k], gh[1 + sp.nd + k]
                gz[1 + sp.nd + k] = d1 * ghkk
                gz[1 + k] += 2.0 * d2 * zk * ghkk
                acc = acc + (d3 * zk * zk + d2 * zkk) * ghkk
        for k in range(sp.n3):                                          # adjoint of the third-order jet (the derivatives of
            z1, z2, z3, g3 = z[1 + k], z[1 + sp.nd + k], z[sp.i3(k)], gh[sp.i3(k)]     # the activation depend on z0 as well)
            gz[sp.i3(k)] = d1 * g3
            gz[1 + sp.nd + k] += 3.0 * d2 * z1 * g3
            gz[1 + k] += (3.0 * d3 * z1 * z1 + 3.0 * d2 * z2) * g3
            acc = acc + (d4 * z1 ** 3 + 3.0 * d3 * z1 * z2 + d2 * z3) * g3
        gz[0] = acc
        for k_, (src, dst, dst_pre, src_pre) in enumerate(sp.skips):
            if src == l and src_pre:
                gz = gz + gskip[k_]
        for k_, (src, dst, dst_pre, src_pre) in enumerate(sp.skips):
            if dst == l and dst_pre:
                gskip[k_] = gz
        db[l] = gz[0].sum(axis=0)
        if l > 0:
            dW[l] = np.einsum('sno,snk->ok', gz, h_prev)
            gh = np.einsum('sno,ok->snk', gz, sp.W[l])
        else:
            g = gz[0].T @ xs                                             # value stream: z0 = W x + b
            for k, c in enumerate(sp.dir_cols):
                for j in _cols(c):
                    g[:, j] += gz[1 + k].sum(axis=0)                     # z_k = sum of W[:, col] over the direction
            dW[0] = g
    return dW, db


def _bc_factors(sp, xs):
    """ P and, per direction, dP/dx_c and d2P/dx_c2 (zero unless the direction is a spatial column). """
    N = xs.shape[0]
    p = np.ones((sp.nsp, N)); p1 = np.zeros((sp.nsp, N)); p2 = np.zeros((sp.nsp, N))
    for j in range(sp.nsp):
        lo, hi = sp.domain[j]
        w = hi - lo
        p[j] = ((xs[:, j] - lo) / w) * ((hi - xs[:, j]) / w)
        p1[j] = (lo + hi - 2.0 * xs[:, j]) / (w * w)
        p2[j] = -2.0 / (w * w)
    P = np.prod(p, axis=0) if sp.nsp else np.ones(N)
    Pk = np.zeros((sp.nd, N)); Pkk = np.zeros((sp.nd, N))
    def rest(*skip):
        out = np.ones(N)
        for j in range(sp.nsp):
            if j not in skip:
                out = out * p[j]
        return out
    for k, c in enumerate(sp.dir_cols):
        sp_cols = [j for j in _cols(c) if j < sp.nsp]
        for j in sp_cols:
            Pk[k] += p1[j] * rest(j)
            Pkk[k] += p2[j] * rest(j)
        if len(sp_cols) == 2:                                            # diagonal direction: cross term 2 P_ab
            Pkk[k] += 2.0 * p1[sp_cols[0]] * p1[sp_cols[1]] * rest(*sp_cols)
    return P, Pk, Pkk


def _ic_gate(sp, xs):
    """ G = sigmoid(tau) - 1/2, its t-derivatives per direction and its log_scale-derivatives. """
    N = xs.shape[0]
    tcol = sp.ndims - 1
    t0 = sp.domain[-1][0]
    es = np.exp(-sp.log_scale)
    tau = (xs[:, tcol] - t0) * es
    s, d1, d2, d3, d4 = act_derivs(tau, 'sigmoid', fourth=True)
    G = s - 0.5
    Gk = np.zeros((sp.nd, N)); Gkk = np.zeros((sp.nd, N)); Gkkk = np.zeros((sp.nd, N))
    dG_ds = -tau * d1
    dGk_ds = np.zeros((sp.nd, N)); dGkk_ds = np.zeros((sp.nd, N)); dGkkk_ds = np.zeros((sp.nd, N))
    for k, c in enumerate(sp.dir_cols):
        if tcol in _cols(c):
            Gk[k] = d1 * es
            Gkk[k] = d2 * es * es
            Gkkk[k] = d3 * es ** 3
            dGk_ds[k] = es * (-tau * d2 - d1)
            dGkk_ds[k] = es * es * (-tau * d3 - 2.0 * d2)
            dGkkk_ds[k] = es ** 3 * (-tau * d4 - 3.0 * d3)          # d tau / ds = -tau, d es / ds = -es
    return G, Gk, Gkk, dG_ds, dGk_ds, dGkk_ds, Gkkk, dGkkk_ds


def ansatz_forward(sp, net, xs, ic_streams=None):
    """ net streams [S,N] -> u streams [S,N] (reference model_torch.py:107-128 + product rule). """
    nd, n2, n3 = sp.nd, sp.n2, sp.n3
    Q = net.copy()
    bc = None
    if sp.has_bc:
        P, Pk, Pkk = _bc_factors(sp, xs)
        bc = (P, Pk, Pkk)
        Q[0] = net[0] * P + sp.bc_value
        for k in range(nd):
            Q[1 + k] = net[1 + k] * P + net[0] * Pk[k]
        for k in range(n2):
            Q[1 + nd + k] = net[1 + nd + k] * P + 2.0 * net[1 + k] * Pk[k] + net[0] * Pkk[k]
        for k in range(n3):                          # single columns: every factor of P is quadratic, P''' = 0
            Q[sp.i3(k)] = net[sp.i3(k)] * P + 3.0 * (net[1 + nd + k] * Pk[k] + net[1 + k] * Pkk[k])
    u = Q.copy()
    gate = None
    if sp.has_ic:
        gate = _ic_gate(sp, xs)
        G, Gk, Gkk = gate[:3]
        Gkkk = gate[6]
        u[0] = G * Q[0]
        for k in range(nd):
            u[1 + k] = Gk[k] * Q[0] + G * Q[1 + k]
        for k in range(n2):
            u[1 + nd + k] = Gkk[k] * Q[0] + 2.0 * Gk[k] * Q[1 + k] + G * Q[1 + nd + k]
        for k in range(n3):                          # u''' = G Q''' + 3 G' Q'' + 3 G'' Q' + G''' Q
            u[sp.i3(k)] = G * Q[sp.i3(k)] + 3.0 * (Gk[k] * Q[1 + nd + k] + Gkk[k] * Q[1 + k]) + Gkkk[k] * Q[0]
        if ic_streams is not None:
            u = u + ic_streams
    return u, (net, Q, bc, gate)


def ansatz_backward(sp, gu, cache):
    """ gu [S,N] -> (gnet [S,N], d log_scale). """
    net, Q, bc, gate = cache
    nd, n2, n3 = sp.nd, sp.n2, sp.n3
    gQ = gu.copy()
    g_ls = 0.0
    if sp.has_ic:
        G, Gk, Gkk, dG, dGk, dGkk, Gkkk, dGkkk = gate
        gG = gu[0] * Q[0]
        gQ[0] = gu[0] * G
        for k in range(nd):
            gG = gG + gu[1 + k] * Q[1 + k]
            gGk = gu[1 + k] * Q[0]
            gQ[0] = gQ[0] + gu[1 + k] * Gk[k]
            gQ[1 + k] = gu[1 + k] * G
            if k < n2:
                gkk = gu[1 + nd + k]
                gG = gG + gkk * Q[1 + nd + k]
                gGk = gGk + 2.0 * gkk * Q[1 + k]
                g_ls += np.sum(gkk * Q[0] * dGkk[k])
                gQ[0] = gQ[0] + gkk * Gkk[k]
                gQ[1 + k] = gQ[1 + k] + 2.0 * gkk * Gk[k]
                gQ[1 + nd + k] = gkk * G
            g_ls += np.sum(gGk * dGk[k])
        for k in range(n3):
            g3 = gu[sp.i3(k)]
            gQ[sp.i3(k)] = g3 * G
            gQ[1 + nd + k] = gQ[1 + nd + k] + 3.0 * g3 * Gk[k]
            gQ[1 + k] = gQ[1 + k] + 3.0 * g3 * Gkk[k]
            gQ[0] = gQ[0] + g3 * Gkkk[k]
            gG = gG + g3 * Q[sp.i3(k)]
            g_ls += np.sum(g3 * (3.0 * (Q[1 + nd + k] * dGk[k] + Q[1 + k] * dGkk[k]) + Q[0] * dGkkk[k]))
        g_ls += np.sum(gG * dG)
    gnet = gQ.copy()
    if sp.has_bc:
        P, Pk, Pkk = bc
        gnet[0] = gQ[0] * P
        for k in range(nd):
            gnet[0] = gnet[0] + gQ[1 + k] * Pk[k]
            gnet[1 + k] = gQ[1 + k] * P
            if k < n2:
                gkk = gQ[1 + nd + k]
                gnet[0] = gnet[0] + gkk * Pkk[k]
                gnet[1 + k] = gnet[1 + k] + 2.0 * gkk * Pk[k]
                gnet[1 + nd + k] = gkk * P
        for k in range(n3):
            g3 = gQ[sp.i3(k)]
            gnet[sp.i3(k)] = g3 * P
            gnet[1 + nd + k] = gnet[1 + nd + k] + 3.0 * g3 * Pk[k]
            gnet[1 + k] = gnet[1 + k] + 3.0 * g3 * Pkk[k]
    return gnet, g_ls


def step(sp, xs, residual, ic_streams=None, n_global=None):
    """ One residual + grad evaluation.  `residual(u_streams, xs) -> (r [N], dr/du_streams [S,N])`.
    Returns dict(u, r, loss, dW, db, dlog_scale, u_streams). """
    xs = np.asarray(xs, dtype=np.float64)
    n_global = n_global or xs.shape[0]
    net, fcache = mlp_jet_forward(sp, xs)
    u, acache = ansatz_forward(sp, net, xs, ic_streams)
    r, dr_du = residual(u, xs)
    loss = np.sum(r * r) / n_global
    gu = dr_du * (2.0 * r / n_global)[None, :]
    gnet, g_ls = ansatz_backward(sp, gu, acache)
    dW, db = mlp_jet_backward(sp, gnet, fcache)
    return dict(u=u[0], r=r, loss=loss, dW=dW, db=db, dlog_scale=g_ls, u_streams=u, gu=gu)


def adam_update(p, g, m, v, step_no, lr=0.005, b1=0.9, b2=0.999, eps=1e-8):
    """ torch.optim.Adam single-tensor form (defaults: no weight decay, no amsgrad). """
    m[:] = b1 * m + (1 - b1) * g
    v[:] = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step_no
    bc2 = 1 - b2 ** step_no
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    p[:] = p - (lr / bc1) * m / denom


def act_d5(z, act):
    """ fifth derivative of the activation (the reverse sweep of fourth-order streams needs it; round 5) -- the formulas of
    pinn_kernel.h pinn_act_d5 in fp64; tests/test_activations.py holds them to torch's nested autograd. """
    z = np.asarray(z, dtype=np.float64)
    act, par = act_param(act)
    if act == 'softplus' and par is not None and par != 1.0:
        return par ** 4 * act_d5(par * z, 'softplus')
    if act == 'tanh':
        t2 = np.tanh(z) ** 2
        return (1.0 - t2) * (16.0 - 120.0 * t2 + 120.0 * t2 * t2)
    if act == 'sin':
        return np.cos(z)
    if act in ('identity', 'relu', 'leakyrelu'):
        return np.zeros_like(z)
    if act in ('elu', 'selu'):
        s_, al = (1.0507009873554805, 1.6732632423543772) if act == 'selu' else (1.0, 1.0 if par is None else par)
        return np.where(z > 0, 0.0, s_ * al * np.exp(z))
    if act == 'softsign':
        return 120.0 / (1.0 + np.abs(z)) ** 6
    if act == 'gelu':
        phi = np.exp(-0.5 * z * z) / np.sqrt(2.0 * np.pi)
        return phi * z * ((z * z - 11.0) * z * z + 18.0)
    sg = 1.0 / (1.0 + np.exp(-z))
    a, q = sg * (1.0 - sg), 1.0 - 2.0 * sg
    s4, s5 = a * q * (q * q - 8.0 * a), a * (q ** 4 - 22.0 * a * q * q + 16.0 * a * a)
    if act == 'sigmoid':
        return s5
    if act == 'softplus':
        return s4
    if act in ('silu', 'swish'):
        return z * s5 + 5.0 * s4
    if act == 'logsigmoid':
        return -s4
    if act == 'tanhshrink':
        u, u1, u2, u3, u4, u5 = z, 1.0, 0.0, 0.0, 0.0, 0.0
    elif act == 'gelu_tanh':
        k, c = 0.7978845608028654, 0.044715
        u, u1, u2, u3, u4, u5 = k * (z + c * z ** 3), k * (1.0 + 3.0 * c * z * z), 6.0 * k * c * z, 6.0 * k * c, 0.0, 0.0
    elif act == 'mish':
        u, u1, u2, u3, u4, u5 = np.logaddexp(0.0, z), sg, a, a * q, a * (q * q - 2.0 * a), s4
    else:
        raise ValueError(act)
    T = np.tanh(u)
    T2 = T * T
    a1 = 1.0 - T2
    a2, a3, a4, a5 = -2.0 * T * a1, a1 * (6.0 * T2 - 2.0), a1 * T * (16.0 - 24.0 * T2), a1 * (16.0 - 120.0 * T2 + 120.0 * T2 * T2)
    t4 = a4 * u1 ** 4 + 6.0 * a3 * u1 ** 2 * u2 + a2 * (4.0 * u1 * u3 + 3.0 * u2 ** 2) + a1 * u4
    t5 = a5 * u1 ** 5 + 10.0 * a4 * u1 ** 3 * u2 + a3 * (15.0 * u1 * u2 ** 2 + 10.0 * u1 ** 2 * u3) + a2 * (10.0 * u2 * u3 + 5.0 * u1 * u4) + a1 * u5
    if act == 'tanhshrink':
        return -t5
    return (0.5 if act == 'gelu_tanh' else 1.0) * (z * t5 + 5.0 * t4)
