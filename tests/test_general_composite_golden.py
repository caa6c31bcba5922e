"""GeneralSolver (diffusion / BSDE loss, unbounded domains): the package's composite plan against the
reference's fixed-seed runs (tests/golden/*, produced by the reference's own GeneralSolver.train)."""
import math

import pytest
import torch

from conftest import load_golden
from util_cases import psp

CASES = ["dwgen_d10_diffusion", "dwgen_d10_bsde", "allencahn_d10_diffusion", "heat_d6_diffusion",
         "dwgen_d7_h20_diffusion", "allencahn_d20_default_diffusion", "dwgen_d40_h50_bsde"]


def build(case, device="cpu", backend="auto"):
    prob = getattr(psp, case["problem"]["kind"])(device=device, **case["problem"]["kwargs"])
    model = psp.GeneralSolver(problem=prob, name=case["name"], verbose=False, device=device, backend=backend,
                              **case["solver"])
    if "net" in case:
        model.V = psp.DenseNet(d_in=prob.d + 1, d_out=1, lr=case["solver"]["lr"], arch=case["net"]["arch"],
                               seed=case["net"]["seed"]).to(device)
    return prob, model


@pytest.mark.parametrize("name", CASES)
def test_general_composite_matches_reference(name):
    rec = load_golden(name)
    exact = rec["torch"] == torch.__version__
    torch.set_num_threads(1)
    prob, model = build(rec["case"])
    model.train()
    assert model.plan_name == "torch"
    exp = rec["expected"]
    assert model.K_log == exp["K_log"]
    for got, want in zip(model.loss_log, exp["loss_log"]):
        assert (got == want) if exact else math.isclose(got, want, rel_tol=1e-5)
    xp = torch.tensor(exp["probe_x"]).reshape(-1, prob.d)
    tp = torch.full((xp.shape[0], 1), exp["probe_t"])
    with torch.no_grad():
        v = model.V(torch.cat([xp, tp], 1)).squeeze()
    want = torch.tensor(exp["probe_V"])
    assert torch.allclose(v, want, rtol=1e-5, atol=1e-7)


def test_out_of_scope_variants_raise():
    rec = load_golden("heat_d6_diffusion")
    prob, model = build(rec["case"])
    model.loss_method = "PINN"
    with pytest.raises(NotImplementedError):
        model.train()
