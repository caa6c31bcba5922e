"""The split-product backward of the wide family for d <= 256 (csrc/hjbwx_kernels.h: producers form dz2 and dz1 on split W3^T / W2^T
tables, consumers contract pairs of sample blocks in three v_mfma_f32_16x16x32_f16 per weight-gradient tile) against the fp32-MFMA
kernel it replaces (PSP_WIDE_BWD_X3 = 0, hjbw_bwd2_kernel) on the same path store: wide instances with d <= 256 (padded ones included: d = 129, 160), ragged K (a
last sample block with few trajectories, a last round with surplus blocks), block counts that are odd (d = 200: the 16x16x16 MFMA
tail) and even, both loss modes, and run-to-run bit equality (the tail once read a register pair while it was still being written).
The forward is the same kernel in both runs, so D must be bit-equal and only the gradient is compared."""
import os

import pytest
import torch

from util_cases import psp

pytestmark = pytest.mark.gpu


def _run(x3, problem, K, N, dev, loss="log-variance", seed=42):
    old = {k: os.environ.get(k) for k in ("PSP_WIDE_BWD_X3",)}
    os.environ["PSP_WIDE_BWD_X3"] = x3
    try:
        m = psp.Solver("wx", problem, lr=1e-3, L=1, K=K, delta_t=0.01, loss_method=loss, time_approx="inner",
                       adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=seed,
                       device=dev, backend="native", noise="philox", widths=(64, 64))
        m.train()
        assert m.plan_name == "native"
        plan = m._native_plan
        return plan.D.cpu().clone(), plan.grad.cpu().clone(), list(m.loss_log)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("d,K,N", [(200, 1003, 7), (192, 96, 5), (256, 250, 6), (129, 77, 4), (160, 515, 9), (200, 16 * 1024 + 5, 3)])
def test_split_backward_equals_fp32_mfma_backward(d, K, N):
    dev = torch.device("cuda:0")
    prob = psp.LLGC(d=d, off_diag=0.1 / d ** 0.5, T=N * 0.01, seed=42, device=dev)
    D0, g0, l0 = _run("0", prob, K, N, dev)
    D1, g1, l1 = _run("1", prob, K, N, dev)
    assert torch.equal(D0, D1) and l0 == l1                               # same forward, same loss
    gmax = float(g0.abs().max())
    assert gmax > 0
    # fp32-grade split products: observed ~1e-6 of the gradient's scale; the bound is the parity suite's
    assert float((g1 - g0).abs().max()) <= 2e-5 * gmax
    _, g2, _ = _run("1", prob, K, N, dev)
    assert torch.equal(g1, g2)                                            # run-to-run bit equality
    assert not torch.equal(g1, g0)                                        # the two switches do select different kernels


def test_split_backward_other_losses_and_problems():
    dev = torch.device("cuda:0")
    d = 200
    for prob, loss in ((psp.DoubleWell_multidim(d=d, d_1=3, d_2=d - 3, T=0.06, eta=0.5, kappa=2.0, device=dev), "log-variance"),
                       (psp.LLGC(d=d, off_diag=0.0, T=0.06, seed=42, device=dev), "variance")):
        D0, g0, l0 = _run("0", prob, 300, 6, dev, loss=loss)
        D1, g1, l1 = _run("1", prob, 300, 6, dev, loss=loss)
        assert torch.equal(D0, D1)
        assert float((g1 - g0).abs().max()) <= 2e-5 * float(g0.abs().max())


def test_split_backward_survives_a_large_weight_spread():
    """The trajectory weights are scaled by one power of two per launch (from a scan of D); a batch whose weights span many
    binades (a long horizon: D of very different sizes) still gives the fp32 kernel's gradient."""
    dev = torch.device("cuda:0")
    prob = psp.LLGC(d=200, off_diag=0.05, T=1.0, seed=42, device=dev)
    D0, g0, _ = _run("0", prob, 512, 100, dev)
    D1, g1, _ = _run("1", prob, 512, 100, dev)
    assert torch.equal(D0, D1)
    assert float(D0.abs().max()) / max(1e-30, float(D0.abs().min())) > 1e2
    assert float((g1 - g0).abs().max()) <= 2e-5 * float(g0.abs().max())


@pytest.mark.parametrize("d,K,N", [(320, 2048, 12), (384, 1040, 20), (448, 1100, 18), (500, 2064, 10), (512, 1500, 16)])
def test_streaming_backward_above_256_against_fp32_on_the_same_store(d, K, N):
    """hjbw_bwd_x3_kernel (d > 256: four waves, the split W3^T table resident in LDS, ONE exchange buffer per workgroup -- a wave
    that runs ahead into the next round writes it while the others are still in the last phase of the current one) against the
    fp32-MFMA backward on the same path store with caller-supplied trajectory weights: several rounds per workgroup, a ragged
    last tile, per workgroup and in total; three runs bit-equal."""
    from test_gpu_range_guard import _bwd, nat
    from util_cases import flat_params
    dev = torch.device("cuda:0")
    prob = psp.LLGC(d=d, off_diag=0.1 / d ** 0.5, T=N * 0.01, seed=42, device=dev)
    m = psp.Solver("wx", prob, lr=1e-3, L=1, K=K, delta_t=0.01, loss_method="log-variance", time_approx="inner",
                   adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42,
                   device=dev, backend="native", noise="philox", widths=(64, 64), mlp_dtype="f16x3")
    params0 = flat_params(m.z_n).to(dev)
    m.train()
    plan = m._native_plan
    assert plan.matrix_mode == "f16x3" and plan.family == 2
    g = torch.Generator(device="cpu").manual_seed(11)
    w = (torch.randn(K, generator=g) * (2.0 / K)).to(dev)
    g3, rows3 = _bwd(plan, m, params0, w, nat.MLP_F16X3)
    g2, rows2 = _bwd(plan, m, params0, w, nat.MLP_FP32)
    assert rows3.shape[0] * 4 < N * ((K + 15) // 16), "every workgroup should run more than one round"
    scale = float(g2.abs().max())
    assert torch.isfinite(g3).all() and not torch.equal(g3, g2)
    assert float((g3 - g2).abs().max()) <= 2e-5 * scale
    rs = rows2.abs().max(dim=1).values.clamp_min(1e-30)
    assert float(((rows3 - rows2).abs().max(dim=1).values / rs).max()) <= 1e-4
    for _ in range(2):
        g3b, rows3b = _bwd(plan, m, params0, w, nat.MLP_F16X3)
        assert torch.equal(g3b, g3) and torch.equal(rows3b, rows3)
