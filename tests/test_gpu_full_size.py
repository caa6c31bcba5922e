"""Full-size and many-round tests of the wide (hjbw_kernels.h) and DenseNet-control (hjbd_kernels.h) kernel families.

At BASELINE.json's sizes the oracle is out of reach, so these use size-independent properties (determinism, shard
independence through k_offset, fp64 sum consistency, finite gradient); next to them the oracle is compared on d = 200 and
d = 500 problems that are large enough for the PERSISTENT structure of the kernels to matter (several rounds per workgroup,
two workgroups per CU at d = 200), on supplied noise.
"""
import math

import pytest
import torch

from util_cases import orc, psp

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _solver(d, K, T, name="full", **over):
    prob = psp.LLGC(d=d, off_diag=0.1 / d ** 0.5, T=T, seed=42, device=dev())
    kw = dict(lr=1e-3, L=1, K=K, delta_t=0.01, loss_method="log-variance", time_approx="inner",
              adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False,
              seed=42, device=dev(), backend="native", noise="philox", widths=(64, 64))
    kw.update(over)
    return prob, psp.Solver(name, prob, **kw), kw


# BASELINE.json configs[3] per-GPU share (d=200, K=32768, N=100) and configs[4] shape at the resident-store size
@pytest.mark.parametrize("d,K,T,N", [(200, 32768, 1.0, 100), (500, 16384, 2.0, 200)])
def test_wide_family_full_size_properties(d, K, T, N):
    prob, a, kw = _solver(d, K, T)
    a.train()
    prob_b, b, _ = _solver(d, K, T)
    b.train()
    assert a.N == N and a._native_plan.family == 2
    pa, pb = a._native_plan, b._native_plan
    # (a) determinism
    assert a.loss_log == b.loss_log and math.isfinite(a.loss_log[0])
    assert torch.equal(pa.D, pb.D) and torch.equal(pa.grad, pb.grad)
    assert bool(torch.isfinite(pa.grad).all()) and float(pa.grad.abs().max()) > 0.0
    # (b) shard independence: the upper half alone, k_offset = K/2, reproduces the full run's D bit for bit
    half = psp.Solver("h", prob, **dict(kw, K=K // 2))
    plan = psp.plan_native.HjbNativePlan(half, noise="philox")
    plan.cfg.k_offset = K // 2
    losses = torch.zeros(1, device=dev())
    plan.iteration(0, losses)
    torch.cuda.synchronize()
    assert torch.equal(plan.D, pa.D[K // 2:])
    # (c) the kernel's fp64 partial sums against a plain fp64 sum of D
    tot, tot2 = pa.D.double().sum(), (pa.D.double() ** 2).sum()
    assert abs(float(pa.sums[0] - tot)) <= 1e-9 * float(pa.D.double().abs().sum()) + 1e-9
    assert abs(float(pa.sums[1] - tot2)) <= 1e-9 * float(tot2) + 1e-9
    # (d) the gradient is linear in the trajectory weights: the chunked 'two_gradient' plan recombines two backward
    #     launches per chunk into the same gradient
    prob_c, c, _ = _solver(d, K, T, path_chunks=4)
    c.train()
    pc = c._native_plan
    assert pc.n_chunks == 4 and torch.equal(pc.D, pa.D)
    assert float((pc.grad - pa.grad).abs().max()) <= 1e-5 * float(pa.grad.abs().max())


@pytest.mark.parametrize("d,K,N", [(200, 8192, 16), (500, 8192, 8)])
def test_wide_family_many_rounds_match_oracle(d, K, N):
    """N * K / 64 rounds over at most 512 (d=200) / 256 (d=500) persistent workgroups: >= 4 rounds each, ragged last round
    excluded by construction; D and the flat gradient against the oracle's autograd on the reference's noise stream."""
    T = N * 0.01 + 0.005
    prob, model, kw = _solver(d, K, T, noise="reference")
    assert model.N == N
    model.train()
    plan = model._native_plan
    assert plan.family == 2
    nround = (N * ((K + 15) // 16) + 3) // 4
    assert nround >= 4 * plan.sizes.bwd_workgroups, (nround, plan.sizes.bwd_workgroups)
    torch.set_num_threads(16)
    oprob = orc.make_problem("LLGC", d=d, off_diag=0.1 / d ** 0.5, T=T, seed=42)
    ocfg = orc.HJBConfig(K=K, delta_t=0.01, lr=1e-3, L=1, seed=42, adaptive_forward_process=True, detach_forward=True)
    z = orc.TanhMLP(d + 1, d, 1e-3, seed=123, widths=(64, 64))
    _, y0, oN = orc.hjb_build(oprob, ocfg)
    ref = orc.hjb_train(oprob, ocfg, step_models=(z, y0, oN), trace=True)
    tr = ref["traces"][0]
    D, D_ref = plan.D.cpu(), tr["D"]
    scale = max(1.0, float(D_ref.abs().max()))
    assert float((D - D_ref).abs().max()) <= 2e-5 * scale
    g, g_ref = plan.grad.cpu(), torch.cat([t.reshape(-1) for t in tr["grads"]])
    err = float((g - g_ref).abs().max()) / float(g_ref.abs().max())
    print("d=%d K=%d N=%d: %d rounds on %d workgroups, gradient rel err %.1e" % (d, K, N, nround, plan.sizes.bwd_workgroups, err))
    assert err <= 2e-4
    cond = float((D_ref.double() ** 2).mean()) / max(abs(ref["loss_log"][0]), 1e-30)
    assert math.isclose(model.loss_log[0], ref["loss_log"][0], rel_tol=min(1e-4, max(2e-5, 4 * 6e-8 * cond)))


def test_outer_dense_control_full_size_properties():
    """time_approx='outer' (the reference's constructor default) at d=100, K=65536, N=50: one DenseNet per time step."""
    d, K = 100, 65536
    prob = psp.LLGC(d=d, off_diag=0.01, T=0.5, seed=42, device=dev())
    kw = dict(lr=1e-3, L=1, K=K, delta_t=0.01, loss_method="log-variance", time_approx="outer",
              adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False,
              seed=42, device=dev(), backend="native", noise="philox")
    a = psp.Solver("a", prob, **kw)
    a.train()
    b = psp.Solver("b", prob, **kw)
    b.train()
    assert a.N == 50 and a.plan_name == "native"
    pa, pb = a._native_plan, b._native_plan
    assert isinstance(pa, psp.plan_dense_native.DenseNativePlan) and pa.kernel_bwd
    assert a.loss_log == b.loss_log and math.isfinite(a.loss_log[0])
    assert torch.equal(pa.D, pb.D) and torch.equal(pa.grad, pb.grad)
    assert bool(torch.isfinite(pa.grad).all()) and float(pa.grad.abs().max()) > 0.0
    half = psp.Solver("h", prob, **dict(kw, K=K // 2))
    plan = psp.plan_dense_native.DenseNativePlan(half, noise="philox")
    plan.cfg.base.k_offset = K // 2
    losses = torch.zeros(1, device=dev())
    plan.iteration(0, losses)
    torch.cuda.synchronize()
    assert torch.equal(plan.D, pa.D[K // 2:])
    tot = pa.D.double().sum()
    assert abs(float(pa.sums[0] - tot)) <= 1e-9 * float(pa.D.double().abs().sum()) + 1e-9
