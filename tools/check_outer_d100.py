import sys, torch
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import path_space_pde_solver_amd as psp
from oracle import pathspace_oracle as orc
dev = torch.device('cuda:0')
for K in (512, 8192):
    prob = psp.LLGC(d=100, off_diag=0.01, T=1.0, seed=42, device=dev)
    m = psp.Solver('chk', prob, lr=1e-3, L=3, K=K, delta_t=0.01, loss_method='log-variance', time_approx='outer',
                   adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42, device=dev,
                   backend='native', noise='reference')
    m.train()
    torch.set_num_threads(16)
    oprob = orc.make_problem('LLGC', d=100, off_diag=0.01, T=1.0, seed=42)
    cfg = orc.HJBConfig(K=K, delta_t=0.01, lr=1e-3, L=3, seed=42, loss_method='log-variance', time_approx='outer', adaptive_forward_process=True, detach_forward=True)
    out = orc.hjb_train(oprob, cfg, step_models=orc.hjb_build(oprob, cfg))
    print(K, 'native', m.loss_log, 'oracle', out['loss_log'], 'Dmax', float(m._native_plan.D.abs().max()))
