import sys, time, torch
sys.path.insert(0, '.')
import path_space_pde_solver_amd as psp
dev = torch.device('cuda:0')
prob = psp.LLGC(d=200, off_diag=0.1 / 200 ** 0.5, T=1.0, seed=42, device=dev)
m = psp.Solver('att', prob, lr=1e-3, L=12, K=32768, delta_t=0.01, loss_method='log-variance', time_approx='inner',
               adaptive_forward_process=True, detach_forward=False, u_l2_error_flag=False, verbose=False, seed=42,
               device=dev, backend='native', noise='philox', widths=(64, 64), mlp_dtype=(sys.argv[1] if len(sys.argv) > 1 else 'auto'))
plan = m._choose_plan()
losses = torch.zeros(12, device=dev)
for l in range(3):
    plan.iteration(l, losses)
torch.cuda.synchronize(); t0 = time.perf_counter()
for l in range(3, 11):
    plan.iteration(l, losses)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 8
print('d=200 attached: %.2f ms/iteration = %.3g trajectory-timesteps/s' % (1e3 * el, 32768 * 100 / el))
