"""The cooperative forward of the wide family (csrc/hjbc_kernels.h: 2 or 4 trajectory tiles per 512-thread workgroup, the output
blocks of the d x d products dealt out over its eight waves) computes the same rollout as the tile-per-wave kernel it replaces:
forced with PSP_FWD_COOP = 2 / 4 it is compared with PSP_FWD_COOP = 0 on every dimension class it serves -- two blocks per wave
with an odd block count (d = 200: a padding block), an even one (d = 192, 256), four blocks per wave with idle waves (d = 320) and
without (d = 500) -- on ragged K (a last tile with few trajectories, a last workgroup with surplus tiles), for dense and
element-wise drift / sigma kinds and both process modes.  The backward kernels read the
path store either forward leaves, so the gradient is part of every comparison."""
import os

import pytest
import torch

from util_cases import psp

pytestmark = pytest.mark.gpu


def _run(coop, problem, K, N, d, dev, noise="philox", adaptive=True, seed=42, L=1):
    old = os.environ.get("PSP_FWD_COOP")
    os.environ["PSP_FWD_COOP"] = coop
    try:
        m = psp.Solver("coop", problem, lr=1e-3, L=L, K=K, delta_t=0.01, loss_method="log-variance", time_approx="inner",
                       adaptive_forward_process=adaptive, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=seed,
                       device=dev, backend="native", noise=noise, widths=(64, 64))
        m.train()
        assert m.plan_name == "native"
        plan = m._native_plan
        tiles = int(getattr(plan.sizes, "fwd_coop_tiles", 0))
        return plan.D.cpu().clone(), plan.grad.cpu().clone(), list(m.loss_log), tiles
    finally:
        if old is None:
            os.environ.pop("PSP_FWD_COOP", None)
        else:
            os.environ["PSP_FWD_COOP"] = old


@pytest.mark.parametrize("d,K", [(200, 1003), (192, 96), (256, 250), (320, 133), (500, 173)])
@pytest.mark.parametrize("tiles", ["2", "4"])
def test_cooperative_forward_equals_tile_per_wave(d, K, tiles):
    dev = torch.device("cuda:0")
    prob = psp.LLGC(d=d, off_diag=0.1 / d ** 0.5, T=0.07, seed=42, device=dev)
    D0, g0, l0, t0 = _run("0", prob, K, 7, d, dev)
    D1, g1, l1, t1 = _run(tiles, prob, K, 7, d, dev)
    assert t0 == 0 and t1 == int(tiles), (t0, t1)
    scale = max(1.0, float(D0.abs().max()))
    # same products, same order per element; the row sums |Z|^2, Z.xi and the terminal cost are combined over eight waves
    assert float((D1 - D0).abs().max()) <= 2e-6 * scale
    assert float((g1 - g0).abs().max()) <= 2e-5 * float(g0.abs().max())
    assert abs(l1[0] - l0[0]) <= 1e-5 * abs(l0[0])
    Dr, gr, lr, _ = _run(tiles, prob, K, 7, d, dev)                       # deterministic
    assert torch.equal(Dr, D1) and torch.equal(gr, g1)


@pytest.mark.parametrize("kind", ["dw", "lqgc_like_diag", "nonadaptive"])
def test_cooperative_forward_other_coefficient_kinds(kind):
    """Element-wise drift (double well), diagonal drift with identity sigma, and the non-adaptive process (stored image
    xi + sqrt(dt) Z): the branches of the step that are not products."""
    dev = torch.device("cuda:0")
    d = 200
    if kind == "dw":
        prob = psp.DoubleWell_multidim(d=d, d_1=3, d_2=d - 3, T=0.06, eta=0.5, kappa=2.0, device=dev)
    else:
        prob = psp.LLGC(d=d, off_diag=0.0, T=0.06, seed=42, device=dev)
    adaptive = kind != "nonadaptive"
    D0, g0, l0, _ = _run("0", prob, 300, 6, d, dev, adaptive=adaptive)
    for tiles in ("2", "4"):
        D1, g1, l1, t1 = _run(tiles, prob, 300, 6, d, dev, adaptive=adaptive)
        assert t1 == int(tiles)
        assert float((D1 - D0).abs().max()) <= 2e-6 * max(1.0, float(D0.abs().max()))
        assert float((g1 - g0).abs().max()) <= 2e-5 * float(g0.abs().max())


def test_supplied_noise_keeps_the_tile_per_wave_kernel():
    """The cooperative kernel generates its increments on the device; reference noise (the parity mode, pinned on the goldens by
    tests/test_gpu_parity.py) stays on hjbw_fwd_kernel whatever the switch says -- bit for bit the same run."""
    dev = torch.device("cuda:0")
    prob = psp.LLGC(d=200, off_diag=0.01, T=0.05, seed=42, device=dev)
    D, g, l, tiles = _run("4", prob, 64, 5, 200, dev, noise="reference")
    D0, g0, l0, t0 = _run("0", prob, 64, 5, 200, dev, noise="reference")
    assert tiles == 0 and t0 == 0
    assert torch.equal(D, D0) and torch.equal(g, g0) and l == l0


def test_philox_rollout_statistics_match_between_the_kernels_at_scale():
    """K = 16 384 at d = 500 (the default selection: four tiles per workgroup): loss and gradient of the two kernels on the same
    counters, one full-size launch each."""
    dev = torch.device("cuda:0")
    prob = psp.LLGC(d=500, off_diag=0.1 / 500 ** 0.5, T=0.05, seed=42, device=dev)
    D0, g0, l0, t0 = _run("0", prob, 16384, 5, 500, dev)
    old = os.environ.pop("PSP_FWD_COOP", None)
    try:
        m = psp.Solver("coop", prob, lr=1e-3, L=1, K=16384, delta_t=0.01, loss_method="log-variance", time_approx="inner",
                       adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42,
                       device=dev, backend="native", noise="philox", widths=(64, 64))
        m.train()
        plan = m._native_plan
        assert int(plan.sizes.fwd_coop_tiles) == 4
        assert float((plan.D.cpu() - D0).abs().max()) <= 2e-6 * max(1.0, float(D0.abs().max()))
        assert float((plan.grad.cpu() - g0).abs().max()) <= 2e-5 * float(g0.abs().max())
    finally:
        if old is not None:
            os.environ["PSP_FWD_COOP"] = old
