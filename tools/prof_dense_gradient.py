import sys, torch
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
import path_space_pde_solver_amd as psp
dev = torch.device('cuda:0')
d, K, N, H = 100, 65536, 100, 30
prob = psp.LLGC(d=d, off_diag=0.01, T=N * 0.01, seed=42, device=dev)
m = psp.Solver(name='t', problem=prob, loss_method='log-variance', time_approx='outer', L=2, lr=1e-3, seed=42, delta_t=0.01, K=K,
               adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, device=dev, backend='native', noise='philox')
m.z_n = [psp.DenseNet(d_in=d, d_out=d, lr=1e-3, arch=[H, H], seed=42).to(dev) for _ in range(m.N)]
m.update_Phis()
m.train()
plan = m._native_plan
w = torch.randn(K, device=dev)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        plan._gradient(w)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
