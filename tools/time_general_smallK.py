import os, sys, time
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
import torch
import path_space_pde_solver_amd as psp
dev = torch.device("cuda:0")
for K in (200, 1024, 4096):
    prob = psp.DoubleWell_multidim_for_general_solver(d=100, d_1=50, d_2=50, T=0.3, eta=1.0, kappa=1.0, modus="HJB", device=dev)
    m = psp.GeneralSolver(prob, "seg", seed=42, delta_t=0.001, N=100, lr=1e-3, L=30, K=K, K_boundary=50, loss_method="diffusion",
                          verbose=False, device=dev, backend="native", noise="philox")
    m.V = psp.DenseNet(d_in=101, d_out=1, lr=1e-3, arch=[64, 64], seed=42).to(dev)
    plan = m._choose_plan()
    for l in range(5):
        plan.iteration(l)
    torch.cuda.synchronize()
    plan.events = []
    t0 = time.perf_counter(); n = 20
    for l in range(5, 5 + n):
        plan.iteration(l)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    ev = plan.events
    fwd = sum(e[0].elapsed_time(e[1]) for e in ev) / n
    bwd = sum(e[2].elapsed_time(e[3]) for e in ev) / n
    print("GeneralSolver d=100 N=100 K=%d: %.3f ms per iteration (forward kernel %.3f, backward kernel %.3f)" % (K, wall, fwd, bwd))
