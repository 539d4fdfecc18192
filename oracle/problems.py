""" TEST INFRASTRUCTURE ONLY -- stream-form statements of the BASELINE residuals for `oracle.jet_f64`.

For each config: which input columns are differentiated (`dir_cols`, second-order directions first), how many
of them need a second derivative (`n2`), and `residual(u_streams [S,N], xs [N,d]) -> (r [N], dr/du [S,N])`.
The pydens-form callables live in `pinn_configs.py`; tests check both forms agree through the oracle.
"""
import numpy as np

PI = np.pi


def stream_form(name):
    if name in ('cfg1', 'cfg2'):                     # u_xx + u_yy - 5 sin(pi (x+y)); streams u,ux,uy,uxx,uyy
        def residual(u, xs):
            r = u[3] + u[4] - 5.0 * np.sin(PI * (xs[:, 0] + xs[:, 1]))
            d = np.zeros_like(u); d[3] = 1.0; d[4] = 1.0
            return r, d
        return dict(dir_cols=[0, 1], n2=2, residual=residual)
    if name == 'cfg3':                               # u_xx + u_yy - u_t; dirs x,y (2nd) then t (1st): u,ux,uy,ut,uxx,uyy
        def residual(u, xs):
            r = u[4] + u[5] - u[3]
            d = np.zeros_like(u); d[4] = 1.0; d[5] = 1.0; d[3] = -1.0
            return r, d
        return dict(dir_cols=[0, 1, 2], n2=2, residual=residual)
    if name in ('cfg4', 'ode_sigmoid'):              # u_x - e pi cos(e pi x); streams u,ux
        def residual(u, xs):
            x, e = xs[:, 0], xs[:, 1]
            r = u[1] - e * PI * np.cos(e * PI * x)
            d = np.zeros_like(u); d[1] = 1.0
            return r, d
        return dict(dir_cols=[0], n2=0, residual=residual)
    if name == 'cfg5':                               # u_tt - u_xx; dirs x(col 0), t(col 1): u,ux,ut,uxx,utt
        def residual(u, xs):
            r = u[4] - u[3]
            d = np.zeros_like(u); d[4] = 1.0; d[3] = -1.0
            return r, d
        return dict(dir_cols=[0, 1], n2=2, residual=residual)
    if name == 'mixed':                              # u_xx + u_xy + 2 u_yy - sin(3xy); dirs x, y, (x,y) diagonal v:
        def residual(u, xs):                         # streams u,ux,uy,uv,uxx,uyy,uvv ; u_xy = (uvv - uxx - uyy) / 2
            r = 0.5 * u[4] + 1.5 * u[5] + 0.5 * u[6] - np.sin(3.0 * xs[:, 0] * xs[:, 1])
            d = np.zeros_like(u); d[4] = 0.5; d[5] = 1.5; d[6] = 0.5
            return r, d
        return dict(dir_cols=[0, 1, (0, 1)], n2=3, residual=residual)
    if name == 'heat3d':                             # u_xx + u_yy + u_zz - u_t; dirs x,y,z (2nd) then t (1st):
        def residual(u, xs):                         # streams u,ux,uy,uz,ut,uxx,uyy,uzz
            r = u[5] + u[6] + u[7] - u[4]
            d = np.zeros_like(u); d[5] = 1.0; d[6] = 1.0; d[7] = 1.0; d[4] = -1.0
            return r, d
        return dict(dir_cols=[0, 1, 2, 3], n2=3, residual=residual)
    if name == 'kdv':                                # u_t + 6 u u_x + u_xxx; dirs x (3rd order), t (1st):
        def residual(u, xs):                         # streams u, ux, ut, uxx, uxxx
            r = u[2] + 6.0 * u[0] * u[1] + u[4]
            d = np.zeros_like(u); d[2] = 1.0; d[0] = 6.0 * u[1]; d[1] = 6.0 * u[0]; d[4] = 1.0
            return r, d
        return dict(dir_cols=[0, 1], n2=1, n3=1, residual=residual)
    if name == 'resnet3':                            # u_t + u u_x + 0.05 u_xxx - 0.1 u_xx; dirs x (3rd order), t (1st):
        def residual(u, xs):                         # streams u, ux, ut, uxx, uxxx
            r = u[2] + u[0] * u[1] + 0.05 * u[4] - 0.1 * u[3]
            d = np.zeros_like(u); d[2] = 1.0; d[0] = u[1]; d[1] = u[0]; d[4] = 0.05; d[3] = -0.1
            return r, d
        return dict(dir_cols=[0, 1], n2=1, n3=1, residual=residual)
    raise KeyError(name)


def ic_streams_f64(name, xs, dir_cols, n2, n3=0):
    """ IC(x_spatial) and its derivative streams in fp64 (analytic, per config); None when IC is constant-free. """
    xs = np.asarray(xs, dtype=np.float64)
    nd = len(dir_cols)
    S = 1 + nd + n2 + n3
    out = np.zeros((S, xs.shape[0]))
    if name in ('kdv', 'resnet3'):                   # x sin(pi x): streams u, ux, ut, uxx, uxxx
        x = xs[:, 0]
        sn, cs = np.sin(PI * x), np.cos(PI * x)
        out[0] = x * sn
        out[1] = sn + PI * x * cs
        out[3] = 2 * PI * cs - PI * PI * x * sn
        out[4] = -3 * PI * PI * sn - PI ** 3 * x * cs
        return out
    if name == 'cfg3':                               # 10 x y (1-x)(1-y)
        x, y = xs[:, 0], xs[:, 1]
        fx, fy = x * (1 - x), y * (1 - y)
        out[0] = 10 * fx * fy
        out[1] = 10 * (1 - 2 * x) * fy; out[2] = 10 * fx * (1 - 2 * y)
        out[4] = -20 * fy; out[5] = -20 * fx
        return out
    if name == 'heat3d':                             # 8 x y z (1-x)(1-y)(1-z)
        x, y, z = xs[:, 0], xs[:, 1], xs[:, 2]
        fx, fy, fz = x * (1 - x), y * (1 - y), z * (1 - z)
        out[0] = 8 * fx * fy * fz
        out[1] = 8 * (1 - 2 * x) * fy * fz; out[2] = 8 * fx * (1 - 2 * y) * fz; out[3] = 8 * fx * fy * (1 - 2 * z)
        out[5] = -16 * fy * fz; out[6] = -16 * fx * fz; out[7] = -16 * fx * fy
        return out
    if name == 'cfg4':
        out[0] = 1.0
        return out
    if name == 'ode_sigmoid':
        out[0] = 2.0
        return out
    if name == 'cfg5':                               # x (1-x)
        x = xs[:, 0]
        out[0] = x * (1 - x); out[1] = 1 - 2 * x; out[3] = -2.0
        return out
    return None
