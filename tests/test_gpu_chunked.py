"""K-chunked native plan (plan_native.py: path stores larger than a budget) against the resident-store plan.

The reference keeps the whole autograd graph of an iteration (solver.py:420-557), so any K that fits its memory runs; the
native plan bounds its path store by processing the trajectories in chunks.  Both chunk modes must give the same D, loss
and gradient as the unchunked iteration (SURVEY.md 8e parity criterion style: equal up to fp32 summation order):
  * per-trajectory D_k            bit-identical (same kernel, same Philox counters / supplied noise)
  * loss                          <= 1e-6 relative
  * flat gradient                 <= 1e-6 * max|g| for 'recompute', <= 5e-6 * max|g| for 'two_gradient'
and the chunked run must still reproduce the reference's golden loss_log (<= 1e-4).
"""
import math

import pytest
import torch

from conftest import load_golden
from util_cases import make_pkg_solver

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


# (golden case, chunk modes that apply)
CASES = [("llgc_d100_h64_logvar", ("two_gradient", "recompute")),
         ("lqgc_d2_logvar_noul2", ("two_gradient", "recompute")),
         ("llgc_d200_h64_logvar", ("two_gradient", "recompute")),          # wide family
         ("llgc_d300_h40_logvar", ("two_gradient", "recompute")),          # wide family, d > 256: hjbw_bwd_x3_kernel under
         ("llgc_d500_h64_logvar", ("two_gradient", "recompute")),          #   PSP_LOSS_WEIGHTS (the path configs[4] takes)
         ("lqgc_d33_h50_logvar", ("two_gradient", "recompute")),           # padded instance
         ("lqgc_d2_moment", ("two_gradient", "recompute")),                # learn_Y_0
         ("lqgc_d4_randx0", ("two_gradient", "recompute")),
         ("llgc_d8_logvar_ul2", ("two_gradient", "recompute")),
         ("lqgc_d2_variance", ("recompute",)),
         ("lqgc_d2_variance_learn_y0", ("recompute",)),
         ("lqgc_d2_cross_entropy", ("recompute",)),
         ("llgc_d100_h64_attached_logvar", ("recompute",)),
         ("dw_d10_attached_moment", ("recompute",)),
         ("lqgc_d4_relative_entropy", ("recompute",)),
         ("llgc_d20_relative_entropy_detached", ("recompute",)),
         ("lqgc_d2_attached_cross_entropy", ("recompute",))]


def _run(case, L, noise, **over):
    model = make_pkg_solver(case, dev(), backend="native", noise=noise, L=L, **over)
    model.train()
    assert model.plan_name == "native"
    return model, model._native_plan


@pytest.mark.parametrize("name,modes", CASES)
@pytest.mark.parametrize("noise", ["reference", "philox"])
def test_chunked_iteration_equals_resident_store(name, modes, noise):
    case = load_golden(name)["case"]
    base, bplan = _run(case, 1, noise)
    assert bplan.n_chunks == 1
    D0, g0 = bplan.D.clone(), bplan.grad.clone()
    for mode in modes:
        for chunks in (3, 4):
            m, plan = _run(case, 1, noise, path_chunks=chunks, chunk_mode=mode)
            assert plan.n_chunks >= 2 and plan.chunk_mode == mode, (plan.n_chunks, plan.chunk_mode)
            assert torch.equal(plan.D, D0), (name, mode, float((plan.D - D0).abs().max()))
            assert math.isclose(m.loss_log[0], base.loss_log[0], rel_tol=1e-6), (m.loss_log, base.loss_log)
            err = float((plan.grad - g0).abs().max()) / float(g0.abs().max())
            print("%s %s %s chunks=%d: gradient rel err %.2e" % (name, noise, mode, plan.n_chunks, err))
            assert err <= (5e-6 if mode == "two_gradient" else 1e-6), (name, mode, err)
            if m.u_L2_loss and base.u_L2_loss and base.u_L2_loss[0] != 0.0:
                assert math.isclose(m.u_L2_loss[0], base.u_L2_loss[0], rel_tol=1e-6)


@pytest.mark.parametrize("name,modes", CASES)
def test_chunked_training_matches_reference_golden(name, modes):
    rec = load_golden(name)
    exp = rec["expected"]
    for mode in modes:
        m, plan = _run(rec["case"], rec["case"]["solver"]["L"], "reference", path_chunks=3, chunk_mode=mode)
        assert plan.n_chunks == 3
        assert len(m.loss_log) == len(exp["loss_log"])
        for l, (got, want) in enumerate(zip(m.loss_log, exp["loss_log"])):
            assert math.isclose(got, want, rel_tol=1e-4), (name, mode, l, m.loss_log, exp["loss_log"])
        for got, want in zip(m.Y_0_log, exp["Y_0_log"]):
            assert math.isclose(got, want, rel_tol=1e-4, abs_tol=1e-6)


def test_budget_selects_the_chunk_count():
    """path_budget_bytes -> number of chunks: the store of one chunk stays under the budget, chunks are whole 16-tiles."""
    case = load_golden("llgc_d100_h64_logvar")["case"]
    import os
    old_variant = os.environ.get("PSP_FWD_VARIANT")
    try:
        # the library picks the forward kernel from the trajectory count of a LAUNCH (K <= 1024: four trajectories per
        # workgroup, K <= 8192: feature split): D is bit-identical between chunked and resident runs of the SAME kernel,
        # and equal to summation order (a few ulp) when the chunk size moves the launch to another one
        os.environ["PSP_FWD_VARIANT"] = "2"
        full, fplan = _run(case, 1, "philox", K=4096)
        store = int(fplan.sizes.path_bytes)
        m, plan = _run(case, 1, "philox", K=4096, path_budget_bytes=store // 5 + 1)
        assert torch.equal(plan.D, fplan.D)
    finally:
        if old_variant is None:
            os.environ.pop("PSP_FWD_VARIANT", None)
        else:
            os.environ["PSP_FWD_VARIANT"] = old_variant
    full, fplan = _run(case, 1, "philox", K=4096)
    m, plan = _run(case, 1, "philox", K=4096, path_budget_bytes=store // 5 + 1)
    assert plan.n_chunks >= 5 and plan.chunk_K % 16 == 0
    assert int(plan.sizes.path_bytes) <= store // 5 + 1 + 16 * store // 4096
    assert plan.path.numel() * 4 == int(plan.sizes.path_bytes)
    assert float((plan.D - fplan.D).abs().max()) <= 2e-6 * max(1.0, float(fplan.D.abs().max()))
    assert math.isclose(m.loss_log[0], full.loss_log[0], rel_tol=1e-6)
    err = float((plan.grad - fplan.grad).abs().max()) / float(fplan.grad.abs().max())
    assert err <= 5e-6, err


def test_ragged_last_chunk_on_another_forward_kernel():
    """the forward grid is not monotone in K_local.  K = 2064 in two chunks on 256 CUs: chunk 0 has 65 tiles
    (feature-split kernel, grid 65), chunk 1 has 64 tiles (quad kernel, grid 256) -- the shared scratch must be sized for
    the larger of the two, per buffer.  fp32 products: the small-K kernels exist in that mode only."""
    case = load_golden("llgc_d100_h64_logvar")["case"]
    K = 2064
    base, bplan = _run(case, 1, "philox", K=K, mlp_dtype="fp32")
    for mode in ("two_gradient", "recompute"):
        m, plan = _run(case, 1, "philox", K=K, mlp_dtype="fp32", path_chunks=2, chunk_mode=mode)
        assert plan.n_chunks == 2 and plan.chunk_K == 1040, (plan.n_chunks, plan.chunk_K)
        from path_space_pde_solver_amd import native as nat
        for off, k, c, c0, cw in plan.chunks:
            q = nat.query(c)
            assert q.fwd_partial_bytes <= plan.fp_stride * 8 and q.grad_partial_bytes <= plan.grad_partial.numel() * 4
            assert q.path_bytes <= plan.path.numel() * 4
        # chunk 0 runs the kernel of the resident launch (bit-identical), chunk 1 another one: equal to summation order
        assert torch.equal(plan.D[:1040], bplan.D[:1040])
        assert float((plan.D - bplan.D).abs().max()) <= 5e-6 * max(1.0, float(bplan.D.abs().max()))
        assert math.isclose(m.loss_log[0], base.loss_log[0], rel_tol=2e-6), (m.loss_log, base.loss_log)
        err = float((plan.grad - bplan.grad).abs().max()) / float(bplan.grad.abs().max())
        assert err <= 2e-5, (mode, err)


def test_two_gradient_is_shift_invariant_under_a_large_mean():
    """The 'two_gradient' combination (2/K)[G1 - (mean D - c) G0] must not lose digits when |mean D| >> std D: a learnable
    Y_0 far from the optimum shifts every D_k by the same constant, which the log-variance gradient ignores."""
    case = load_golden("llgc_d100_h64_logvar")["case"]
    over = dict(K=2048, learn_Y_0=True)
    base, bplan = _run(case, 1, "philox", **over)

    def shifted(**kw):
        model = make_pkg_solver(case, dev(), backend="native", noise="philox", L=1, **over, **kw)
        with torch.no_grad():
            model.y_0.Y_0.fill_(1000.0)
        model.train()
        return model, model._native_plan

    ref, rplan = shifted()
    m, plan = shifted(path_chunks=4, chunk_mode="two_gradient")
    assert plan.n_chunks == 4
    scale = float(rplan.grad.abs().max())
    assert float((plan.grad - rplan.grad).abs().max()) <= 2e-5 * scale
    # and the shift itself leaves the (resident-store) gradient where it was, up to the fp32 resolution of D ~ 1000
    assert float((rplan.grad - bplan.grad).abs().max()) <= 2e-3 * scale


def test_plan_is_rebuilt_when_sizes_change():
    """a plan sized for (K, N, loss) must not be reused after the solver's attributes were changed."""
    case = load_golden("lqgc_d2_logvar_noul2")["case"]
    model = make_pkg_solver(case, dev(), backend="native", noise="philox", L=1)
    model.train()
    p1 = model._native_plan
    model.K = 256
    model.train()
    p2 = model._native_plan
    assert p2 is not p1 and p2.K_local == 256 and p2.D.numel() == 256
    model.loss_method = "moment"
    model.train()
    assert model._native_plan is not p2
