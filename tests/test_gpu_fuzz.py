"""Randomised differential test of the native Solver plan against the CPU oracle: 28 configurations drawn from a fixed
seed over state dimension (1 .. 140), hidden width (3 .. 64), trajectory count (1 .. 300, mostly ragged), step count
(1 .. 9), problem kind, loss, adaptive / detached flags, random initial points and learnable Y_0.  First iteration:
D_k, gradient and loss with the bars of test_gpu_parity.py."""
import math
import random

import pytest
import torch

from util_cases import flat_params, make_oracle, make_pkg_solver, orc

pytestmark = pytest.mark.gpu


def _draw(i):
    rng = random.Random(1000 + i)
    kind = rng.choice(["LLGC", "LLGC", "LQGC", "DoubleWell_multidim"])
    d = rng.choice([1, 2, 3, 5, 9, 16, 17, 31, 48, 65, 100, 113, 140])
    H = rng.choice([3, 8, 16, 17, 30, 33, 48, 64])
    K = rng.choice([1, 2, 15, 16, 17, 33, 64, 100, 257, 300])
    N = rng.choice([1, 2, 3, 5, 9])
    dt = 0.05
    loss = rng.choice(["log-variance", "log-variance", "moment", "variance", "cross_entropy", "relative_entropy"])
    adaptive = rng.random() < 0.75
    detach = rng.random() < 0.6
    if loss == "relative_entropy":
        adaptive = True                      # (the reference's relative entropy assumes the controlled process)
    if not adaptive:
        detach = True                        # nothing to differentiate through when c = 0
    if loss == "variance" and K < 2:
        K = 2                                # unbiased variance of one sample is undefined in the reference too
    if loss == "log-variance" and K == 1:
        K = 3                                # variance of a single D is identically 0: gradient and its tolerance degenerate
    if kind == "DoubleWell_multidim":
        kwargs = dict(d=d, d_1=d // 2, d_2=d - d // 2, T=N * dt, eta=0.05, kappa=0.5)
    elif kind == "LQGC":
        kwargs = dict(d=d, off_diag=0.05, T=N * dt, seed=42, delta_t=dt)
    else:
        kwargs = dict(d=d, off_diag=rng.choice([0.0, 0.3 / d ** 0.5]), T=N * dt, seed=42)
    solver = dict(loss_method=loss, time_approx="inner", adaptive_forward_process=adaptive, detach_forward=detach,
                  early_stopping_time=None, L=1, lr=0.002, seed=42, delta_t=dt, K=K, u_l2_error_flag=False,
                  random_X_0=rng.random() < 0.3, learn_Y_0=(loss == "moment" and rng.random() < 0.5))
    return dict(name="fuzz%d" % i, family="solver", problem=dict(kind=kind, kwargs=kwargs), solver=solver,
                net=dict(kind="tanh_mlp", widths=[H, H], seed=123))


@pytest.mark.parametrize("mode", ["fp32", "f16x3"])          # fp32-MFMA kernels / split-product kernels: same bars
@pytest.mark.parametrize("i", range(28))
def test_random_configuration_matches_oracle(i, mode):
    case = _draw(i)
    s = case["solver"]
    model = make_pkg_solver(case, torch.device("cuda:0"), backend="native", L=1, mlp_dtype=mode)
    oprob, ocfg, omodels = make_oracle(case, L=1)
    assert torch.equal(flat_params(model.z_n), flat_params(omodels[0]))
    model.train()
    assert model.plan_name == "native", (case, model.plan_reason)
    plan = model._native_plan
    ref = orc.hjb_train(oprob, ocfg, step_models=omodels, trace=True)
    tr = ref["traces"][0]
    D_ref = -tr["Zsum_g"] if s["loss_method"] == "relative_entropy" else tr["D"]
    D = plan.D.cpu()
    finite = torch.isfinite(D_ref)
    assert torch.equal(torch.isfinite(D), finite), case       # explicit Euler on the double well can overflow: then in both
    err_D = float((D - D_ref)[finite].abs().max()) if bool(finite.any()) else 0.0
    assert err_D <= 2e-5 * max(1.0, float(D_ref[finite].abs().max()) if bool(finite.any()) else 1.0), case
    if not bool(finite.all()):
        assert not math.isfinite(model.loss_log[0]) and not math.isfinite(ref["loss_log"][0])
        return
    g, g_ref = plan.grad.cpu(), torch.cat([x.reshape(-1) for x in tr["grads"]])
    assert g.shape == g_ref.shape
    gmax = float(g_ref.abs().max())
    assert float((g - g_ref).abs().max()) <= 2e-4 * gmax + 1e-12, (case, float((g - g_ref).abs().max()), gmax)
    lref = ref["loss_log"][0]
    cond = float((D_ref.double() ** 2).mean()) / max(abs(lref), 1e-30)
    tol = min(1e-3, max(2e-5, 4 * 6e-8 * cond))
    if s["loss_method"] in ("variance", "cross_entropy"):      # losses of exp(D): an absolute error in D is a relative one there
        tol = max(tol, 8.0 * err_D + 1e-5)
    assert math.isclose(model.loss_log[0], lref, rel_tol=tol, abs_tol=1e-7), (case, model.loss_log[0], lref)


def _draw_general(i):
    rng = random.Random(5000 + i)
    fam = rng.choice(["general", "general", "general_bounded", "elliptic"])
    d = rng.choice([1, 2, 3, 7, 16, 17, 33, 64, 100])
    H = rng.choice([4, 16, 20, 33, 48, 64])
    K = rng.choice([1, 5, 16, 17, 65, 150])
    N = rng.choice([1, 2, 5, 11])
    dt = rng.choice([0.01, 0.02])
    loss = rng.choice(["diffusion", "diffusion", "BSDE"])
    adaptive = rng.random() < 0.4
    Kb = rng.choice([2, 6, 10])
    solver = dict(seed=42, delta_t=dt, N=N, lr=0.001, L=1, K=K, K_boundary=min(Kb, K), loss_method=loss,
                  adaptive_forward_process=adaptive)
    attrs = None
    if fam == "general":
        kind = rng.choice(["DoubleWell_multidim_for_general_solver", "AllenCahn", "HeatEquation"])
        T = rng.choice([0.5, 1.5]) * N * dt
        if kind.startswith("DoubleWell"):
            kwargs = dict(d=d, d_1=d // 2, d_2=d - d // 2, T=T, eta=0.1, kappa=0.5, modus=rng.choice(["HJB", "linear"]))
        else:
            kwargs = dict(d=d, T=T, seed=42) if kind == "HeatEquation" else dict(d=d, T=T, seed=42, modus="pt")
        solver["alpha"] = [1.0, rng.choice([0.5, 1.0]), 1.0]
    elif fam == "general_bounded":
        kind = rng.choice(["ExponentialOnSphereNonlinearParabolic", "QuadraticOnBox"])
        T = rng.choice([0.5, 1.5]) * N * dt
        if kind == "QuadraticOnBox":
            kwargs = dict(d=d, T=T, X_l=-1.0, X_r=rng.choice([0.7, 1.0]), one_boundary=rng.random() < 0.3, scale=1.0,
                          quad_h=rng.random() < 0.5)
        else:
            kwargs = dict(d=d, T=T, alpha=0.2)
            if loss == "diffusion" and rng.random() < 0.3:
                attrs = dict(boundary_type="Neumann")
        solver["alpha"] = [1.0, 1.0, rng.choice([0.5, 2.0])]
    else:
        kind = rng.choice(["ExponentialOnSphere", "ExponentialOnBallNonlinear", "ExponentialOnBallNonlinearSin", "QuadraticOnBox"])
        if kind == "QuadraticOnBox":
            kwargs = dict(d=d, X_l=-1.0, X_r=1.0, one_boundary=rng.random() < 0.3, parabolic=False, quad_h=rng.random() < 0.5)
        else:
            kwargs = dict(d=d, alpha=0.2)
        solver["alpha"] = [1.0, rng.choice([0.5, 1.0])]
    if d == 1 and kind == "QuadraticOnBox":
        d = 2
        kwargs["d"] = 2                      # (the reference's square boundary sampler needs d >= 2)
    case = dict(name="gfuzz%d" % i, family=fam, problem=dict(kind=kind, kwargs=kwargs), solver=solver, net=dict(arch=[H, H], seed=42))
    if attrs:
        case["problem"]["attrs"] = attrs
    if kind == "QuadraticOnBox":
        case["numpy_seed"] = 9
        solver["K_boundary"] = max(2, 2 * (solver["K_boundary"] // 2))
    return case


@pytest.mark.parametrize("mode", ["fp32", "f16x3"])
@pytest.mark.parametrize("i", range(24))
def test_random_general_configuration_matches_oracle(i, mode):
    """GeneralSolver / EllipticSolver on unbounded, sphere and box domains: native first iteration against the oracle."""
    from test_general_composite_golden import build as build_pkg
    from test_gpu_bounded_elliptic import oracle_run
    case = _draw_general(i)
    prob, model = build_pkg(case, device=torch.device("cuda:0"), backend="native", L=1, mlp_dtype=mode)
    model.train()
    assert model.plan_name == "native", case
    ref = oracle_run(case, 1)
    assert model.K_log == ref["K_log"], case
    assert math.isclose(model.loss_log[0], ref["loss_log"][0], rel_tol=1e-4, abs_tol=1e-7), (case, model.loss_log, ref["loss_log"])
    g_ref = torch.cat([g.reshape(-1) for g in ref["traces"][0]["grads"]])
    g = model._gen_plan.grad.cpu()
    err = float((g - g_ref).abs().max())
    assert err <= 5e-4 * float(g_ref.abs().max()) + 1e-10, (case, err, float(g_ref.abs().max()))
