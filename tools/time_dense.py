"""Times one training iteration of Solver with a DenseNet control on an MI355X: the native plan (hjbd forward kernel +
GEMM gradient) against the composite torch plan (the reference's op sequence with autograd), same configuration.
Usage: python tools/time_dense.py [outer|inner] [d] [K] [N] [H]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_space_pde_solver_amd as psp  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "outer"
d = int(sys.argv[2]) if len(sys.argv) > 2 else 100
K = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
N = int(sys.argv[4]) if len(sys.argv) > 4 else 50
H = int(sys.argv[5]) if len(sys.argv) > 5 else 30
dev = torch.device("cuda:0")
prob = psp.LLGC(d=d, off_diag=0.1 / d ** 0.5, T=N * 0.01, seed=42, device=dev)


def make(backend, L):
    m = psp.Solver(name="t", problem=prob, loss_method="log-variance", time_approx=mode, L=L, lr=1e-3, seed=42,
                   delta_t=0.01, K=K, adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False,
                   verbose=False, device=dev, backend=backend, noise="philox" if backend == "native" else "reference")
    if mode == "outer":
        m.z_n = [psp.DenseNet(d_in=d, d_out=d, lr=1e-3, arch=[H, H], seed=42).to(dev) for _ in range(m.N)]
    else:
        m.z_n = psp.DenseNet(d_in=d + 1, d_out=d, lr=1e-3, arch=[H, H], seed=42).to(dev)
    m.update_Phis()
    return m


for backend, L in (("native", 12), ("torch", 3)):
    m = make(backend, 2)
    m.train()                      # warm-up (plan construction, first launches)
    torch.cuda.synchronize()
    m.L = L
    m.loss_log = []
    t0 = time.time()
    m.train()
    torch.cuda.synchronize()
    per = (time.time() - t0) / L
    extra = ""
    if backend == "native":
        plan = m._native_plan
        plan.events = []
        losses = torch.zeros(4, device=dev)
        for l in range(4):
            plan.iteration(l, losses)
        torch.cuda.synchronize()
        ev = plan.events[-1]
        extra = "  (forward kernels %.2f ms, gradient %.2f ms: %s)" % (ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3]), "hjbd_bwd_kernel" if plan.kernel_bwd else "library GEMMs")
    print("%s  %s d=%d K=%d N=%d H=%d: %.2f ms per iteration = %.3g trajectory-timesteps/s%s"
          % (backend.ljust(6), mode, d, K, m.N, H, per * 1e3, K * m.N / per, extra), flush=True)
