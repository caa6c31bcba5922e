"""GPU parity of the native GeneralSolver plan (diffusion / BSDE loss) against the oracle's autograd
and the reference's golden loss logs.  Tolerances: gradient <= 5e-4 * max|g| (second-order sweep in
fp32 on both sides), loss per iteration <= 2e-4 relative, active-step counts exact."""
import math

import pytest
import torch

from conftest import load_golden
from util_cases import orc, psp

pytestmark = pytest.mark.gpu
CASES = ["dwgen_d10_diffusion", "dwgen_d10_bsde", "allencahn_d10_diffusion", "heat_d6_diffusion"]


def dev():
    return torch.device("cuda:0")


def build(case, **over):
    prob = getattr(psp, case["problem"]["kind"])(device=dev(), **case["problem"]["kwargs"])
    kw = dict(case["solver"])
    kw.update(over)
    model = psp.GeneralSolver(problem=prob, name=case["name"], verbose=False, device=dev(), backend="native", **kw)
    if "net" in case:
        model.V = psp.DenseNet(d_in=prob.d + 1, d_out=1, lr=kw["lr"], arch=case["net"]["arch"],
                               seed=case["net"]["seed"]).to(dev())
    return prob, model


def oracle_run(case, L):
    prob = orc.make_problem(case["problem"]["kind"], **case["problem"]["kwargs"])
    s = case["solver"]
    cfg = orc.GeneralConfig(K=s["K"], N=s["N"], delta_t=s["delta_t"], lr=s["lr"], L=L, seed=s["seed"],
                            K_boundary=s["K_boundary"], alpha=tuple(s["alpha"]), loss_method=s["loss_method"])
    V = orc.general_build(prob, cfg, arch=case["net"]["arch"] if "net" in case else None)
    return orc.general_train(prob, cfg, V=V, trace=True)


@pytest.mark.parametrize("name", CASES)
def test_first_iteration_gradient_matches_oracle(name):
    rec = load_golden(name)
    case = rec["case"]
    prob, model = build(case, L=1)
    model.train()
    assert model.plan_name == "native"
    ref = oracle_run(case, 1)
    assert math.isclose(model.loss_log[0], ref["loss_log"][0], rel_tol=5e-5), (model.loss_log, ref["loss_log"])
    assert model.K_log == ref["K_log"]
    g_ref = torch.cat([g.reshape(-1) for g in ref["traces"][0]["grads"]])
    # the plan keeps the gradient of the last iteration in plan.grad -- rebuild it: train() ran one step
    plan = model._gen_plan
    g = plan.grad.cpu()
    assert g.shape == g_ref.shape
    err = float((g - g_ref).abs().max())
    assert err <= 5e-4 * float(g_ref.abs().max()), (err, float(g_ref.abs().max()))


@pytest.mark.parametrize("name", CASES)
def test_loss_log_matches_reference_golden(name):
    rec = load_golden(name)
    prob, model = build(rec["case"])
    model.train()
    exp = rec["expected"]
    assert model.K_log == exp["K_log"]
    for l, (got, want) in enumerate(zip(model.loss_log, exp["loss_log"])):
        assert math.isclose(got, want, rel_tol=2e-4), (l, model.loss_log, exp["loss_log"])
    xp = torch.tensor(exp["probe_x"]).reshape(-1, prob.d).to(dev())
    tp = torch.full((xp.shape[0], 1), exp["probe_t"], device=dev())
    with torch.no_grad():
        v = model.V(torch.cat([xp, tp], 1)).squeeze().cpu()
    want = torch.tensor(exp["probe_V"])
    assert float((v - want).abs().max()) <= 2e-4 * max(1.0, float(want.abs().max()))
