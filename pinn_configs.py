""" The five BASELINE.json workloads (+ three parity-only extras) in pydens form (equation callable + Solver kwargs + point sampler).

Neutral module: depends on neither the product package nor the oracle. `D` and `torch` are passed in so the
same definitions drive the reference, the oracle and the HIP engine. Column order is pydens' (spatial..., t,
params...) -- reference model_torch.py:111 -- so BASELINE's "(t,x,y)" heat config is (x,y,t) here.
"""
import math
import numpy as np

PI = math.pi


def mlp(depth, width):
    return dict(layout='fa' * depth + 'f', features=[width] * depth + [1], activation='Tanh')


def make_config(name, D, torch, V=None):
    """ -> dict(equation, solver_kwargs, n_points, low, high, dir_cols, n2); V: the framework's trainable-variable token (the
    'program' breadth workload only) """
    if name in ('cfg1', 'cfg2'):
        def equation(f, x, y):                                   # reference README.md:36-37
            return D(D(f, x), x) + D(D(f, y), y) - 5 * torch.sin(PI * (x + y))
        net = (dict(layout='fa fa fa f', features=[10, 12, 15, 1], activation='Tanh') if name == 'cfg1'
               else mlp(4, 64))
        return dict(equation=equation, solver_kwargs=dict(ndims=2, boundary_condition=1, **net),
                    n_points=100 if name == 'cfg1' else 65536, low=[0, 0], high=[1, 1])
    if name == 'cfg3':
        def equation(f, x, y, t):                                # tutorial heat equation with a == 1
            return D(D(f, x), x) + D(D(f, y), y) - D(f, t)
        return dict(equation=equation,
                    solver_kwargs=dict(ndims=3, boundary_condition=0,
                                       initial_condition=lambda x, y: 10 * x * y * (1 - x) * (1 - y), **mlp(5, 128)),
                    n_points=262144, low=[0, 0, 0], high=[1, 1, 1])
    if name == 'cfg4':
        def equation(f, x, e):                                   # reference README.md:78-79
            return D(f, x) - e * PI * torch.cos(e * PI * x)
        return dict(equation=equation, solver_kwargs=dict(ndims=1, nparams=1, initial_condition=1, **mlp(4, 64)),
                    n_points=1048576, low=[0, 1], high=[1, 5])
    if name == 'cfg5':
        def equation(f, x, t):                                   # reference model_torch.py:237-239 (IC docstring)
            return D(D(f, t), t) - D(D(f, x), x)
        return dict(equation=equation,
                    solver_kwargs=dict(ndims=2, boundary_condition=0, initial_condition=lambda x: x * (1 - x),
                                       **mlp(6, 256)),
                    n_points=1048576, low=[0, 0], high=[1, 1])
    if name == 'ode_sigmoid':                                    # tutorial cells 28-31: default net [20,30,1] Sigmoid
        def equation(f, x, e):
            return D(f, x) - e * PI * torch.cos(e * PI * x)
        return dict(equation=equation, solver_kwargs=dict(ndims=1, nparams=1, initial_condition=2.0),
                    n_points=700, low=[0, .5], high=[1, 5.5])
    if name == 'mixed':                                          # anisotropic diffusion tensor [[1, .5], [.5, 2]]:
        def equation(f, x, y):                                   # a mixed partial D(D(f, x), y)
            return D(D(f, x), x) + D(D(f, x), y) + 2 * D(D(f, y), y) - torch.sin(3 * x * y)
        return dict(equation=equation,
                    solver_kwargs=dict(ndims=2, boundary_condition=0.5, layout='fa fa f', features=[24, 24, 1],
                                       activation='Tanh'),
                    n_points=4096, low=[0, 0], high=[1, 1])
    if name == 'heat3d':                                         # heat equation in (x, y, z, t): four differentiation directions
        def equation(f, x, y, z, t):
            return D(D(f, x), x) + D(D(f, y), y) + D(D(f, z), z) - D(f, t)
        return dict(equation=equation,
                    solver_kwargs=dict(ndims=4, boundary_condition=0,
                                       initial_condition=lambda x, y, z: 8 * x * y * z * (1 - x) * (1 - y) * (1 - z),
                                       layout='fa fa fa f', features=[40, 40, 40, 1], activation='Tanh'),
                    n_points=4096, low=[0, 0, 0, 0], high=[1, 1, 1, 1])
    if name == 'kdv':                                            # Korteweg-de Vries: third derivative in x, first in t
        def equation(f, x, t):
            return D(f, t) + 6 * f * D(f, x) + D(D(D(f, x), x), x)
        return dict(equation=equation,
                    solver_kwargs=dict(ndims=2, boundary_condition=0.2, initial_condition=lambda x: torch.sin(PI * x) * x,
                                       layout='fa fa fa f', features=[24, 24, 24, 1], activation='Tanh'),
                    n_points=4096, low=[0, 0], high=[1, 1])
    if name == 'resnet3':
        # breadth fixture (round 3): dispersive Burgers equation (third derivative in x) on a net with two residual blocks -- one
        # joining IN FRONT of its activation ('f+a': act(W h + skip)), one behind ('fa+') -- and activations outside Tanh / Sigmoid
        def equation(f, x, t):
            return D(f, t) + f * D(f, x) + 0.05 * D(D(D(f, x), x), x) - 0.1 * D(D(f, x), x)
        return dict(equation=equation,
                    solver_kwargs=dict(ndims=2, boundary_condition=0.1, initial_condition=lambda x: torch.sin(PI * x) * x,
                                       layout='faR fa f+a R fa fa+ f', features=[24, 24, 24, 24, 24, 1],
                                       activation=['Sin', 'Tanh', 'SiLU', 'Tanh', 'Sigmoid']),
                    n_points=4096, low=[0, 0], high=[1, 1])
    if name == 'nested_acts':
        # breadth fixture (round 5): a skip connection INSIDE a skip connection ('R .. R .. + .. +': the reference's layout letters nest,
        # model_torch.py:142-156) and the torch-default activation forms of the second kernel set; viscous Burgers with IC + BC
        def equation(f, x, t):
            return D(f, t) + f * D(f, x) - 0.05 * D(D(f, x), x)
        return dict(equation=equation,
                    solver_kwargs=dict(ndims=2, boundary_condition=0.1, initial_condition=lambda x: torch.sin(PI * x) * x,
                                       layout='faR faR fa fa+ fa+ fa f', features=[24, 24, 24, 24, 24, 24, 1],
                                       activation=['ELU', 'Mish', 'Softsign', 'SELU', 'LogSigmoid', 'LeakyReLU']),
                    n_points=4096, low=[0, 0], high=[1, 1])
    if name == 'act_params':
        # breadth fixture (round 6): activation module INSTANCES configured away from torch's defaults -- the reference hands them through
        # to the block (model_torch.py:150 "Sequence of callables, str", :164-168) -- LeakyReLU(negative_slope), ELU(alpha), Softplus(beta);
        # viscous Burgers with IC + BC (second derivatives: the parameters enter every derivative order of the jets)
        def equation(f, x, t):
            return D(f, t) + f * D(f, x) - 0.05 * D(D(f, x), x)
        nn = torch.nn
        return dict(equation=equation,
                    solver_kwargs=dict(ndims=2, boundary_condition=0.1, initial_condition=lambda x: torch.sin(PI * x) * x,
                                       layout='fa fa fa fa f', features=[24, 24, 24, 24, 1],
                                       activation=[nn.Softplus(beta=2.0), nn.ELU(alpha=0.5), nn.LeakyReLU(0.2), nn.Softplus(beta=0.5)]),
                    n_points=4096, low=[0, 0], high=[1, 1])
    if name == 'mixed3':
        # breadth fixture (round 5): MIXED third derivatives u_xxy and u_xyy (nested D in any order, model_torch.py:174-178) next to a
        # first derivative in t: third Taylor coefficients along x + y and x - y, polarised
        def equation(f, x, y, t):
            return D(f, t) + D(D(D(f, x), x), y) - 0.5 * D(D(D(f, x), y), y) + f * D(f, x) - 0.1 * D(D(f, y), y)
        return dict(equation=equation,
                    solver_kwargs=dict(ndims=3, boundary_condition=0.2,
                                       initial_condition=lambda x, y: torch.sin(PI * x) * y, layout='fa fa fa f',
                                       features=[24, 24, 24, 1], activation='Tanh'),
                    n_points=4096, low=[0, 0, 0], high=[1, 1, 1])
    if name == 'biharm':
        # breadth fixture (round 5): the biharmonic operator u_xxxx + 2 u_xxyy + u_yyyy -- fourth derivatives, the mixed one from the
        # fourth Taylor coefficients along x + y and x - y
        def equation(f, x, y):
            return (D(D(D(D(f, x), x), x), x) + 2.0 * D(D(D(D(f, x), x), y), y) + D(D(D(D(f, y), y), y), y)
                    - 10.0 * torch.sin(PI * x) * torch.sin(PI * y))
        return dict(equation=equation,
                    solver_kwargs=dict(ndims=2, boundary_condition=0.0, layout='fa fa f', features=[24, 24, 1], activation='Tanh'),
                    n_points=4096, low=[0, 0], high=[1, 1])
    if name == 'mixed31':
        # breadth fixture (round 6): the mixed fourth-order partials u_xxxy and u_xyyy (nested D in any order, model_torch.py:174-178) --
        # fourth Taylor coefficients along x + y, x - y and the WEIGHTED diagonals 2x + y, 2x - y (include/pinn.h PINN_DIR_DOUBLE)
        def equation(f, x, y):
            return (D(D(D(D(f, x), x), x), y) - 0.5 * D(D(D(D(f, y), x), y), y) + f * D(f, x)
                    - 4.0 * torch.cos(PI * x) * torch.sin(PI * y))
        return dict(equation=equation,
                    solver_kwargs=dict(ndims=2, boundary_condition=0.3, layout='fa fa f', features=[24, 24, 1], activation='Tanh'),
                    n_points=4096, low=[0, 0], high=[1, 1])
    # ---- breadth workloads: timed by `bench.py --workload ...`, not BASELINE configs -----------------------
    if name == 'mixed111':
        # breadth fixture (round 6): u_xyz, the partial of THREE different columns (model_torch.py:174-178), all three inside the boundary factor --
        # third Taylor coefficients along x +- y +- z (include/pinn.h PINN_DIR_MINUS_C)
        def equation(f, x, y, z):
            return D(D(D(f, x), y), z) + f * D(f, x) - 0.2 * D(D(f, z), z) - 3.0 * torch.sin(PI * x) * torch.cos(PI * y) * z
        return dict(equation=equation,
                    solver_kwargs=dict(ndims=3, boundary_condition=0.2, layout='fa fa f', features=[24, 24, 1], activation='Tanh'),
                    n_points=4096, low=[0, 0, 0], high=[1, 1, 1])
    if name in ('burgers64', 'heat64', 'poisson512'):           # round 6 breadth workloads
        if name == 'burgers64':                                  # viscous Burgers in (x, t) on the 4 x 64 Tanh net: residual program, IC + BC (PinnShape 2)
            def burgers(f, x, t):
                return D(f, t) + f * D(f, x) - 0.05 * D(D(f, x), x)
            return dict(equation=burgers, solver_kwargs=dict(ndims=2, boundary_condition=0, initial_condition=lambda x: torch.sin(PI * x), **mlp(4, 64)),
                        n_points=65536, low=[0, 0], high=[1, 1])
        if name == 'heat64':                                     # 1-D heat equation with a source, same net: affine residual, IC + BC
            def heat(f, x, t):
                return D(f, t) - 0.3 * D(D(f, x), x) - 2.0 * torch.exp(-t) * torch.sin(PI * x)
            return dict(equation=heat, solver_kwargs=dict(ndims=2, boundary_condition=0, initial_condition=lambda x: torch.sin(PI * x), **mlp(4, 64)),
                        n_points=65536, low=[0, 0], high=[1, 1])
        def poisson512(f, x, y):                                 # BASELINE config 2's problem on a 4 x 512 net: width 512, generic path, one call per direction
            return D(D(f, x), x) + D(D(f, y), y) - 5 * torch.sin(PI * (x + y))
        return dict(equation=poisson512, solver_kwargs=dict(ndims=2, boundary_condition=1, **mlp(4, 512)), n_points=65536, low=[0, 0], high=[1, 1])
    if name in ('skip128', 'skip256', 'sin64', 'sin128', 'gelu256', 'program', 'generic'):
        def poisson(f, x, y):
            return D(D(f, x), x) + D(D(f, y), y) - 5 * torch.sin(PI * (x + y))
        if name == 'skip128':                                    # skip connection 'R ... +' (reference model_torch.py:142-156), width 128
            net = dict(layout='faR fa fa+ fa f', features=[128, 128, 128, 128, 1], activation='Tanh')
            return dict(equation=poisson, solver_kwargs=dict(ndims=2, boundary_condition=1, **net), n_points=65536,
                        low=[0, 0], high=[1, 1])
        if name == 'skip256':                                    # two residual blocks of width 256 (the usual post-activation ResNet form)
            net = dict(layout='faR fa fa+ R fa fa+ f', features=[256, 256, 256, 256, 256, 1], activation='Tanh')
            return dict(equation=poisson, solver_kwargs=dict(ndims=2, boundary_condition=1, **net), n_points=65536,
                        low=[0, 0], high=[1, 1])
        if name == 'sin128':                                     # 4 x 128 'Sin': full breadth kernel + streamed weight gradients
            net = dict(layout='fa fa fa fa f', features=[128, 128, 128, 128, 1], activation='Sin')
            return dict(equation=poisson, solver_kwargs=dict(ndims=2, boundary_condition=1, **net), n_points=65536,
                        low=[0, 0], high=[1, 1])
        if name == 'gelu256':                                    # Burgers (residual program, IC + BC) on a pre-activation residual net, 4 x 256 GELU
            def burgers(f, x, t):
                return D(f, t) + f * D(f, x) - 0.05 * D(D(f, x), x)
            net = dict(layout='fa fRa fa f+a f', features=[256, 256, 256, 256, 1], activation='GELU')
            return dict(equation=burgers, solver_kwargs=dict(ndims=2, boundary_condition=0, initial_condition=lambda x: torch.sin(PI * x), **net),
                        n_points=65536, low=[0, 0], high=[1, 1])
        if name == 'sin64':                                      # 4 x 64 with activation 'Sin'
            net = dict(layout='fa fa fa fa f', features=[64, 64, 64, 64, 1], activation='Sin')
            return dict(equation=poisson, solver_kwargs=dict(ndims=2, boundary_condition=1, **net), n_points=65536,
                        low=[0, 0], high=[1, 1])
        if name == 'program':                                    # non-affine equation with a trainable coefficient: residual program
            def reaction(f, x, y):
                k = V('k', data=torch.Tensor([1.5]))
                return D(D(f, x), x) + D(D(f, y), y) + k * f * f - 5 * torch.sin(PI * (x + y))
            return dict(equation=reaction, solver_kwargs=dict(ndims=2, boundary_condition=1, **mlp(4, 64)), n_points=65536,
                        low=[0, 0], high=[1, 1])
        # 'generic': BASELINE config 2 forced onto the generic step path (pinn_jet_forward -> torch -> pinn_jet_backward)
        return dict(equation=poisson, solver_kwargs=dict(ndims=2, boundary_condition=1, **mlp(4, 64)), n_points=65536,
                    low=[0, 0], high=[1, 1])
    raise KeyError(name)


CONFIG_NAMES = ('cfg1', 'cfg2', 'cfg3', 'cfg4', 'cfg5')


def sample_points(cfg, n, seed, steps=None):
    """ U[low, high)^d fp32 points, [n,d] or [steps,n,d]. """
    rng = np.random.RandomState(seed)
    low, high = np.asarray(cfg['low'], dtype=np.float64), np.asarray(cfg['high'], dtype=np.float64)
    shape = (n, len(low)) if steps is None else (steps, n, len(low))
    return (low + (high - low) * rng.rand(*shape)).astype(np.float32)
