"""The three forward kernels of the narrow family -- tile per wave (hjb_fwd_kernel), feature split over the waves of a
workgroup (hjbs_fwd_kernel, K <= 8192) and four trajectories per workgroup on 4x4x1 MFMAs (hjbq_fwd_kernel, K <= 1024) --
compute the same rollout: each one, forced with PSP_FWD_VARIANT, is held to the oracle / golden tolerances of
test_gpu_parity.py on cases that cover every coefficient kind, loss, noise mode and padded shape, and the three are compared
with one another on a ragged K with on-device noise."""
import math
import os

import pytest
import torch

from conftest import load_golden
from util_cases import make_oracle, make_pkg_solver, orc, psp

pytestmark = pytest.mark.gpu

CASES = ["lqgc_d2_logvar_noul2", "llgc_d100_h64_logvar", "llgc_d7_default_logvar", "lqgc_d33_h50_logvar", "dw_d70_h64_logvar",
         "dw_d10_logvar", "llgc_d20_diag_logvar", "lqgc_d2_moment", "lqgc_d4_randx0", "llgc_d8_nonadaptive",
         "lqgc_d2_variance_learn_y0", "llgc_d8_cross_entropy_nonadaptive", "llgc_d100_h64_attached_logvar",
         "dw_d10_attached_moment", "lqgc_d4_relative_entropy", "llgc_d8_logvar_ul2", "llgc_d40_moment_ul2"]


@pytest.fixture
def variant(request):
    old = os.environ.get("PSP_FWD_VARIANT")
    os.environ["PSP_FWD_VARIANT"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("PSP_FWD_VARIANT", None)
    else:
        os.environ["PSP_FWD_VARIANT"] = old


@pytest.mark.parametrize("variant", ["1", "2", "3"], indirect=True)
@pytest.mark.parametrize("name", CASES)
def test_forced_variant_matches_oracle_and_golden(name, variant):
    rec = load_golden(name)
    case = rec["case"]
    dev = torch.device("cuda:0")
    model = make_pkg_solver(case, dev, backend="native", L=1)
    oprob, ocfg, omodels = make_oracle(case, L=1)
    model.train()
    assert model.plan_name == "native"
    plan = model._native_plan
    ref = orc.hjb_train(oprob, ocfg, step_models=omodels, trace=True)
    tr = ref["traces"][0]
    D = plan.D.cpu()
    D_ref = -tr["Zsum_g"] if case["solver"]["loss_method"] == "relative_entropy" else tr["D"]
    scale = max(1.0, float(D_ref.abs().max()))
    assert float((D - D_ref).abs().max()) <= 2e-5 * scale
    g = plan.grad.cpu()
    g_ref = torch.cat([x.reshape(-1) for x in tr["grads"]])
    assert float((g - g_ref).abs().max()) <= 2e-4 * float(g_ref.abs().max())
    # all logged iterations against the reference's own run
    model = make_pkg_solver(case, dev, backend="native")
    model.train()
    for a, b in zip(model.loss_log, rec["expected"]["loss_log"]):
        assert math.isclose(a, b, rel_tol=1e-4), (name, variant, model.loss_log, rec["expected"]["loss_log"])
    if rec["expected"].get("u_L2_loss") and model.u_l2_error_flag:
        for a, b in zip(model.u_L2_loss, rec["expected"]["u_L2_loss"]):
            assert math.isclose(a, b, rel_tol=1e-4), (name, variant)


def test_variants_agree_on_philox_noise_ragged_K():
    """K = 1003 (a last tile with 11 trajectories, a last quad with 3), on-device noise: D of the three kernels agrees to
    summation order, the gradients (same backward kernel on the three path stores) to 1e-5."""
    dev = torch.device("cuda:0")
    res = {}
    old = os.environ.get("PSP_FWD_VARIANT")
    try:
        for v in ("1", "2", "3"):
            os.environ["PSP_FWD_VARIANT"] = v
            prob = psp.LLGC(d=100, off_diag=0.01, T=0.2, seed=42, device=dev)
            m = psp.Solver("v", prob, lr=1e-3, L=2, K=1003, delta_t=0.01, loss_method="log-variance", time_approx="inner",
                           adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42,
                           device=dev, backend="native", noise="philox", widths=(64, 64))
            m.train()
            assert m.plan_name == "native"
            res[v] = (m._native_plan.D.cpu().clone(), m._native_plan.grad.cpu().clone(), list(m.loss_log))
    finally:
        if old is None:
            os.environ.pop("PSP_FWD_VARIANT", None)
        else:
            os.environ["PSP_FWD_VARIANT"] = old
    D1, g1, l1 = res["1"]
    for v in ("2", "3"):
        D, g, l = res[v]
        assert float((D - D1).abs().max()) <= 1e-5 * max(1.0, float(D1.abs().max())), v
        assert float((g - g1).abs().max()) <= 1e-5 * float(g1.abs().max()), v
        for a, b in zip(l, l1):
            assert math.isclose(a, b, rel_tol=1e-5), (v, l, l1)
