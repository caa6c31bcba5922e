"""Times one training iteration with gradients through the state path (detach_forward=False) at the headline shape."""
import sys, time, torch
sys.path.insert(0, '.')
import path_space_pde_solver_amd as psp
dev = torch.device('cuda:0')
for loss, mode in (('log-variance', 'fp32'), ('log-variance', 'f16x3'), ('relative_entropy', 'fp32'), ('relative_entropy', 'f16x3')):
    prob = psp.LLGC(d=100, off_diag=0.01, T=1.0, seed=42, device=dev)
    m = psp.Solver('att', prob, lr=1e-3, L=12, K=65536, delta_t=0.01, loss_method=loss, time_approx='inner',
                   adaptive_forward_process=True, detach_forward=False, u_l2_error_flag=False, verbose=False, seed=42,
                   device=dev, backend='native', noise='philox', widths=(64, 64), mlp_dtype=mode)
    plan = m._choose_plan()
    losses = torch.zeros(12, device=dev)
    for l in range(3):
        plan.iteration(l, losses)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for l in range(3, 11):
        plan.iteration(l, losses)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 8
    print('%s attached, matrix products %s: %.2f ms/iteration = %.3g trajectory-timesteps/s; losses %s' % (loss, mode, 1e3 * el, 65536 * 100 / el, [round(float(x), 5) for x in losses[:3].tolist()]))
