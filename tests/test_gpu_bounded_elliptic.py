"""GPU parity of the native plan on bounded domains (sphere / box exit tests inside the forward kernel, Dirichlet /
Neumann terms, BSDE with boundary data) and of EllipticSolver (same kernels, no time input) against the oracle's
autograd and the reference's golden runs.  Tolerances as in test_gpu_general.py: gradient <= 5e-4 * max|g|,
loss per iteration <= 1e-4 relative (BASELINE.json's bar; observed errors are printed), active-step counts exact."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_general_composite_golden import BOUNDED, COMPOSITE_ONLY, ELLIPTIC, R4_BOUNDED, R4_ELLIPTIC, build as build_pkg
from util_cases import general_oracle_run, orc, psp

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def oracle_run(case, L):
    return general_oracle_run(case, L=L, trace=True)[1]


def _two_spheres(case):
    return case["problem"]["kind"] == "Committor" or case["problem"].get("attrs", {}).get("boundary") == "two_spheres"


# round 4: 'two_spheres' (annulus exit test, a batch size that changes every iteration), 'square-corner', the BSDE loss with a
# Neumann boundary, loss_with_stopped / K_test_log / sample_center, the committor notebook's tanh^2 net (BSDE with N = 1500:
# the tiles leave the time loop after a few hundred steps) -- all on the HIP kernels
NATIVE_R4 = COMPOSITE_ONLY + R4_BOUNDED + R4_ELLIPTIC


@pytest.mark.parametrize("name", BOUNDED + ELLIPTIC + NATIVE_R4)
def test_first_iteration_gradient_matches_oracle(name):
    case = load_golden(name)["case"]
    prob, model = build_pkg(case, device=dev(), backend="native", L=1)
    model.train()
    assert model.plan_name == "native"
    ref = oracle_run(case, 1)
    assert model.K_log == ref["K_log"]
    assert math.isclose(model.loss_log[0], ref["loss_log"][0], rel_tol=5e-5), (model.loss_log, ref["loss_log"])
    g_ref = torch.cat([g.reshape(-1) for g in ref["traces"][0]["grads"]])
    g = model._gen_plan.grad.cpu()
    assert g.shape == g_ref.shape
    err = float((g - g_ref).abs().max())
    assert err <= 5e-4 * float(g_ref.abs().max()), (err, float(g_ref.abs().max()))


@pytest.mark.parametrize("name", BOUNDED + ELLIPTIC + NATIVE_R4)
def test_loss_log_matches_reference_golden(name):
    """Several iterations: also checks that the host consumed exactly the reference's number of noise draws
    (the all-stopped break of solver.py:1093-1097 / :742-744), otherwise iteration 2 would see other noise."""
    rec = load_golden(name)
    prob, model = build_pkg(rec["case"], device=dev(), backend="native")
    model.train()
    exp = rec["expected"]
    assert model.K_log == exp["K_log"]
    want_log = exp["loss_log"]
    if _two_spheres(rec["case"]):
        # The committor data g = [|x| > a] is evaluated ON the sampled inner sphere (solver.py:685, problems.py:1569-1570): half of
        # the boundary batch sits at |x| = a up to the rounding of its own normalisation, and whether sqrt(sum(x^2)) lands above a
        # is decided by the LAST BIT of a CPU reduction -- which differs between CPU models (the fixture was generated in the build
        # container; this box's host gives other bits for 1 - 3 of those points per iteration, 1 / K_boundary of the loss each).
        # The package evaluates g on the host copy of the batch, i.e. exactly what the reference computes ON THIS MACHINE: the
        # several-iteration log is therefore compared with the oracle run here (pinned bit for bit on the fixture where the
        # fixture was made), the fixture itself only for the step counts and to the size of such flips.
        want_log = oracle_run(rec["case"], len(exp["loss_log"]))["loss_log"]
        Kb = rec["case"]["solver"]["K_boundary"]
        for got, want in zip(model.loss_log, exp["loss_log"]):
            assert abs(got - want) <= 4.0 / Kb * rec["case"]["solver"].get("alpha", [1.0, 1.0])[1] + 1e-3 * abs(want)
    errs = [abs(got - want) / abs(want) for got, want in zip(model.loss_log, want_log)]
    print("%s: loss rel err per iteration vs the reference %s" % (name, ["%.1e" % e for e in errs]))
    for l, (got, want) in enumerate(zip(model.loss_log, want_log)):
        assert math.isclose(got, want, rel_tol=1e-4), (l, model.loss_log, want_log)     # BASELINE.json: 1e-4
    assert len(model.times) == len(model.loss_log)
    if rec["case"]["family"] == "elliptic":
        # V_L2 log of EllipticSolver.train (solver.py:718, 738, 813): decoded from the X_n images of the path store
        assert len(model.V_L2_log) == len(exp["V_L2_log"])
        for got, want in zip(model.V_L2_log, exp["V_L2_log"]):
            assert math.isclose(got, want, rel_tol=5e-2 if _two_spheres(rec["case"]) else 1e-4, abs_tol=1e-9), (model.V_L2_log, exp["V_L2_log"])
    if exp.get("V_test_L2"):                                  # K_test_log: the same fresh points after every update
        assert len(model.V_test_L2) == len(exp["V_test_L2"])
        for got, want in zip(model.V_test_L2, exp["V_test_L2"]):
            assert math.isclose(got, want, rel_tol=2e-3 if _two_spheres(rec["case"]) else 1e-4), (model.V_test_L2, exp["V_test_L2"])
    xp = torch.tensor(exp["probe_x"]).reshape(-1, prob.d).to(dev())
    if rec["case"]["family"] != "elliptic":
        xp = torch.cat([xp, torch.full((xp.shape[0], 1), exp["probe_t"], device=dev())], 1)
    with torch.no_grad():
        v = model.V(xp).squeeze().cpu()
    want = torch.tensor(exp["probe_V"])
    # (a flipped boundary label changes the SIGN pattern of one Adam step: lr per parameter and iteration at most)
    assert float((v - want).abs().max()) <= (2e-2 if _two_spheres(rec["case"]) else 2e-4) * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("kind", ["sphere", "box"])
def test_exit_times_philox_large(kind):
    """Size-independent properties at K = 2^17 with device noise: determinism (bitwise), every trajectory's frozen
    point is consistent with its exit test, and the active-step count equals sum_k round((t_N - t_0)/dt)."""
    K, N, d = 1 << 17, 60, 12
    if kind == "sphere":
        prob = psp.ExponentialOnSphereNonlinearParabolic(d=d, T=1.0, alpha=0.2, device=dev())
    else:
        prob = psp.QuadraticOnBox(d=d, T=1.0, X_l=-1.0, X_r=1.0, scale=1.0, device=dev())

    def make():
        m = psp.GeneralSolver(problem=prob, name="big", seed=42, delta_t=0.005, N=N, lr=1e-3, L=1, K=K, K_boundary=50,
                              alpha=[1.0, 1.0, 1.0], loss_method="diffusion", verbose=False, device=dev(),
                              backend="native", noise="philox")
        m.V = psp.DenseNet(d_in=d + 1, d_out=1, lr=1e-3, arch=[32, 32], seed=42).to(dev())
        return m

    a, b = make(), make()
    np.random.seed(1)          # GeneralSolver.train leaves numpy unseeded (the square boundary sample shuffles with it)
    a.train()
    np.random.seed(1)
    b.train()
    pa, pb = a._gen_plan, b._gen_plan
    assert a.loss_log == b.loss_log and math.isfinite(a.loss_log[0]) and a.K_log == b.K_log
    assert torch.equal(pa.grad, pb.grad) and torch.equal(pa.YN, pb.YN) and bool(torch.isfinite(pa.grad).all())
    X0 = pa._sample_domain_device(0)
    t0 = torch.rand(K, generator=pa._gen, device=dev()) * prob.T
    steps = torch.round((pa.tN - t0) / pa.cfg.dt)
    assert int(steps.sum().item()) == a.K_log[0]
    assert 0 < a.K_log[0] < K * N
    XN = pa.XN
    if kind == "sphere":
        # a trajectory that stopped early by leaving the ball sits outside it (the test reads the state before the move)
        early = (steps < N) & ((pa.tN + pa.cfg.dt) <= prob.T)
        assert int(early.sum()) > 0
        assert bool((XN[early].norm(dim=1) >= prob.boundary_distance).all())
        assert bool((XN[~early & (steps == N)].norm(dim=1) < 10.0).all())
    else:
        # the box test reads the proposal: nobody ever steps outside
        assert bool(((XN >= prob.X_l) & (XN <= prob.X_r)).all())
        assert int((steps < N).sum()) > 0


def _case(family, kind, kwargs, attrs=None, net=None, **solver):
    c = dict(name="bsweep", family=family, problem=dict(kind=kind, kwargs=kwargs), solver=dict(seed=42, lr=0.001, L=1, **solver))
    if attrs:
        c["problem"]["attrs"] = attrs
    if net:
        c["net"] = dict(arch=net, seed=42)
    if kind == "QuadraticOnBox":
        c["numpy_seed"] = 5
    return c


EDGE_CASES = [
    # ragged K, a single step, K < 16
    _case("general_bounded", "ExponentialOnSphereNonlinearParabolic", dict(d=2, T=0.5, alpha=0.5), delta_t=0.02, N=1, K=37,
          K_boundary=8, alpha=[1.0, 1.0, 1.0], loss_method="diffusion"),
    _case("general_bounded", "QuadraticOnBox", dict(d=3, T=0.4, X_l=-0.7, X_r=0.9, scale=1.5), delta_t=0.02, N=7, K=9,
          K_boundary=6, alpha=[1.0, 2.0, 0.5], loss_method="diffusion"),
    # padded shapes with several state blocks and wide nets; every trajectory exits before N (BSDE needs all stopped)
    _case("general_bounded", "ExponentialOnSphereNonlinearParabolic", dict(d=37, T=0.3, alpha=0.1), net=[50, 50], delta_t=0.01,
          N=40, K=120, K_boundary=10, alpha=[1.0, 1.0, 1.0], loss_method="BSDE"),
    _case("general_bounded", "QuadraticOnBox", dict(d=70, T=0.2, X_l=-1.0, X_r=1.0, scale=0.7), net=[64, 64], delta_t=0.01,
          N=12, K=200, K_boundary=10, alpha=[1.0, 1.0, 1.0], loss_method="diffusion", adaptive_forward_process=True),
    # elliptic: box whose lower bound is positive (the zero padding of the instance lies OUTSIDE the box), one-sided box
    _case("elliptic", "QuadraticOnBox", dict(d=5, X_l=0.5, X_r=2.0, parabolic=False), delta_t=0.01, N=15, K=50, K_boundary=10,
          loss_method="diffusion"),
    _case("elliptic", "QuadraticOnBox", dict(d=18, X_l=-1.0, X_r=0.4, one_boundary=True, parabolic=False, quad_h=False),
          net=[20, 20], delta_t=0.02, N=10, K=33, K_boundary=10, loss_method="diffusion"),
    _case("elliptic", "ExponentialOnBallNonlinear", dict(d=33, alpha=0.1), net=[48, 48], delta_t=0.002, N=25, K=64,
          K_boundary=12, loss_method="diffusion", adaptive_forward_process=True),
]


@pytest.mark.parametrize("case", EDGE_CASES, ids=lambda c: "%s-%s-d%d-K%d" % (c["family"], c["problem"]["kind"][:12],
                                                                         c["problem"]["kwargs"]["d"], c["solver"]["K"]))
def test_edge_shapes_match_oracle(case):
    """Ragged K, single step, padded instances, boxes that exclude the origin: first iteration against the oracle."""
    prob, model = build_pkg(case, device=dev(), backend="native", L=1)
    model.train()
    assert model.plan_name == "native"
    ref = oracle_run(case, 1)
    assert model.K_log == ref["K_log"]
    assert math.isclose(model.loss_log[0], ref["loss_log"][0], rel_tol=5e-5), (model.loss_log, ref["loss_log"])
    g_ref = torch.cat([g.reshape(-1) for g in ref["traces"][0]["grads"]])
    g = model._gen_plan.grad.cpu()
    err = float((g - g_ref).abs().max())
    assert err <= 5e-4 * float(g_ref.abs().max()), (err, float(g_ref.abs().max()))


def test_annulus_philox_large():
    """'two_spheres' with device noise at K_original = 2^16 (the rejection step keeps ~ 1 - (r_1 / r_2)^d of it): bitwise
    determinism, the batch size follows the sampler, every early exit sits outside the annulus, counts consistent."""
    d, N = 6, 80
    prob = psp.Committor(d=d, device=dev())

    def make():
        m = psp.EllipticSolver(problem=prob, name="big", seed=42, delta_t=0.002, N=N, lr=1e-3, L=2, K=1 << 16, K_boundary=50,
                               alpha=[1.0, 1.0], loss_method="diffusion", verbose=False, device=dev(), backend="native",
                               noise="philox", v_l2_error_flag=False)
        m.V = psp.DenseNet_tanh_2(d_in=d, d_out=1, lr=1e-3, arch=[d + 10, d, d, d], seed=42).to(dev())
        return m

    a, b = make(), make()
    a.train()
    b.train()
    pa, pb = a._gen_plan, b._gen_plan
    assert type(pa).__name__ == "GeneralDeepPlan"
    assert a.loss_log == b.loss_log and all(math.isfinite(v) for v in a.loss_log) and a.K_log == b.K_log
    assert torch.equal(pa.grad, pb.grad) and torch.equal(pa.YN, pb.YN) and bool(torch.isfinite(pa.grad).all())
    assert a.K < a.K_original and a.K > 0.9 * a.K_original and pa.YN.numel() == a.K
    steps = torch.round(pa.tN / pa.cfg.dt)
    assert int(steps.sum().item()) == a.K_log[-1] and 0 < a.K_log[-1] < a.K * N
    r = pa.XN.norm(dim=1)
    early = steps < N
    assert int(early.sum()) > 0
    assert bool(((r[early] <= prob.boundary_distance_1) | (r[early] >= prob.boundary_distance_2)).all())
    assert bool(((r[~early] < 3.0)).all())


def test_user_coefficients_keep_the_composite_plan():
    """A problem whose coefficient was replaced on the instance is outside the kernels' catalogue: backend='auto' takes the
    composite torch plan with a warning, backend='native' raises."""
    rec = load_golden("committor_d3_elliptic_diffusion")
    prob, model = build_pkg(rec["case"], device=dev(), backend="auto", L=1)
    prob.h = lambda x, y, z: 0.1 * y
    with pytest.warns(UserWarning, match="composite torch plan"):
        model.train()
    assert model.plan_name == "torch" and model.plan_reason
    prob, model = build_pkg(rec["case"], device=dev(), backend="native", L=1)
    prob.h = lambda x, y, z: 0.1 * y
    with pytest.raises(NotImplementedError):
        model.train()
