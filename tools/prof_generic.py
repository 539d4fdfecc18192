""" Host-side profile (cProfile) of the GENERIC step path (use_fused = False) on the tutorial's variable + constraint
problem: the path is bound by torch dispatch + autograd on the host, not by the kernels. """
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pydens_amd as pa
from pydens_amd import D, V
def odevar(f, x):
    return D(f, x) - 2 * np.pi * torch.cos(2 * np.pi * x) + V('new_var', data=torch.Tensor([1.0]))
solver = pa.Solver(odevar, ndims=1, initial_condition=1, constraints=lambda f, x: f(torch.tensor([0.5])))
terms = ['equation', 'constraint_0']
solver.use_fused = False
solver.fit(niters=20, batch_size=500, lr=0.01, loss_terms=terms)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
solver.fit(niters=300, batch_size=500, lr=0.01, loss_terms=terms)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(8); st.print_callers('_check')
