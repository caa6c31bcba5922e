"""GPU parity of the DenseNet-control plan (csrc/hjbd_kernels.h forward rollout + GEMM gradient, plan_dense_native.py):
time_approx='outer' (the reference's constructor default: one DenseNet(d -> d) per time step) and a
DenseNet(d+1 -> d) swapped into z_n.  Same bars as test_gpu_parity.py: D_k <= 2e-5 max|D|, gradient <= 2e-4 max|g|
against the oracle's autograd, loss logs <= 1e-4 relative against the reference's own runs."""
import math

import pytest
import torch

from conftest import load_golden
from util_cases import flat_params, make_oracle, make_pkg_solver, orc, psp

pytestmark = pytest.mark.gpu
CASES = ["lqgc_d2_outer", "llgc_d100_densenet64_logvar", "llgc_d12_outer_moment", "dw_d20_densenet_nonadaptive", "lqgc_d6_densenet_variance",
         # gradients through the state path (the reference's default flags) and relative entropy: psp_dnet_adjoint_sweep
         "lqgc_d2_outer_attached", "llgc_d12_outer_relative_entropy", "llgc_d12_outer_relative_entropy_detached",
         "dw_d20_densenet_attached_moment", "lqgc_d6_densenet_attached_cross_entropy"]


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", CASES)
def test_first_iteration_D_and_gradient_match_oracle(name):
    case = load_golden(name)["case"]
    model = make_pkg_solver(case, dev(), backend="native", L=1)
    oprob, ocfg, omodels = make_oracle(case, L=1)
    model.train()
    assert model.plan_name == "native" and isinstance(model._native_plan, psp.plan_dense_native.DenseNativePlan)
    plan = model._native_plan
    ref = orc.hjb_train(oprob, ocfg, step_models=omodels, trace=True)
    tr = ref["traces"][0]
    # relative entropy: the kernel's D is -(Zsum + g(X_N)) (include/psp.h)
    D, D_ref = plan.D.cpu(), (-tr["Zsum_g"] if case["solver"]["loss_method"] == "relative_entropy" else tr["D"])
    assert float((D - D_ref).abs().max()) <= 2e-5 * max(1.0, float(D_ref.abs().max()))
    g, g_ref = plan.grad.cpu(), torch.cat([x.reshape(-1) for x in tr["grads"]])
    assert g.shape == g_ref.shape
    assert float((g - g_ref).abs().max()) <= 2e-4 * float(g_ref.abs().max())
    cond = float((D_ref.double() ** 2).mean()) / max(abs(ref["loss_log"][0]), 1e-30)
    tol = min(1e-4, max(2e-5, 4 * 6e-8 * cond))
    assert math.isclose(model.loss_log[0], ref["loss_log"][0], rel_tol=tol), (model.loss_log[0], ref["loss_log"][0])


@pytest.mark.parametrize("name", CASES)
def test_loss_log_matches_reference_golden(name):
    rec = load_golden(name)
    model = make_pkg_solver(rec["case"], dev(), backend="native")
    model.train()
    exp = rec["expected"]
    assert len(model.loss_log) == len(exp["loss_log"])
    for l, (got, want) in enumerate(zip(model.loss_log, exp["loss_log"])):
        assert math.isclose(got, want, rel_tol=1e-4), (l, model.loss_log, exp["loss_log"])
    for got, want in zip(model.Y_0_log, exp["Y_0_log"]):
        assert math.isclose(got, want, rel_tol=1e-4, abs_tol=1e-6)
    if exp["probes"]:
        xp = torch.tensor(exp["probe_x"]).reshape(-1, model.d).to(dev())
        for pr in exp["probes"]:
            with torch.no_grad():
                u = (-model.Z_n(xp, pr["t"])).cpu()
            want = torch.tensor(pr["minus_Z"]).reshape(u.shape)
            assert float((u - want).abs().max()) <= 1e-4 * max(1e-2, float(want.abs().max()))


def test_outer_philox_large_is_deterministic_and_shard_independent():
    """K = 2^15, d = 40, N = 30 with device noise: two runs agree bitwise in D; the upper half of the trajectories run
    alone (k_offset = K/2) reproduces the full run's D."""
    d, K = 40, 1 << 15
    prob = psp.LLGC(d=d, off_diag=0.05, T=0.3, seed=42, device=dev())

    def make(Kx):
        return psp.Solver(name="big", problem=prob, loss_method="log-variance", time_approx="outer", L=1, lr=1e-3, seed=42,
                          delta_t=0.01, K=Kx, adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False,
                          verbose=False, device=dev(), backend="native", noise="philox")

    a, b = make(K), make(K)
    a.train()
    b.train()
    pa, pb = a._native_plan, b._native_plan
    assert a.loss_log == b.loss_log and math.isfinite(a.loss_log[0])
    assert torch.equal(pa.D, pb.D) and bool(torch.isfinite(pa.grad).all())
    half = make(K // 2)
    ph = psp.plan_dense_native.DenseNativePlan(half, noise="philox")
    ph.flat.copy_(torch.cat([p.detach().reshape(-1) for net in make(K).z_n for p in net.W]).to(dev()))
    ph.cfg.base.k_offset = K // 2
    losses = torch.zeros(1, device=dev())
    ph.iteration(0, losses)
    torch.cuda.synchronize()
    assert torch.equal(ph.D, pa.D[K // 2:])


@pytest.mark.parametrize("name", ["llgc_d12_outer_moment", "lqgc_d6_densenet_variance"])
def test_importance_sampling_evaluation_on_the_dense_kernel(name):
    """utilities.do_importance_sampling_me (reference utilities.py:287-359) with a DenseNet control: the forward sweep on
    hjbd_fwd_kernel against the same sweep in torch, same host noise; the evaluation grid (delta_t = 0.01) differs from
    the training grid, so the step -> net / time-feature index map of solver.py:360-362 is exercised."""
    case = load_golden(name)["case"]
    model = make_pkg_solver(case, dev(), backend="native", L=2)
    model.train()
    ut = psp.utilities
    assert ut._native_reason(model.problem, model, "approx", False) is not None      # not the tanh-MLP path
    assert ut._dense_reason(model.problem, model) is None
    torch.manual_seed(11)
    got = ut.do_importance_sampling_me(model.problem, model, 400, delta_t=0.01)
    torch.manual_seed(11)
    want = ut._is_composite(model.problem, model, 400, 0.01)
    for a, b in zip(got, want):
        assert math.isclose(a, b, rel_tol=2e-4), (got, want)


# (time_approx, problem kind, d, H, K, delta_t, T, adaptive): corners of the instance grid, ragged K, single step, tiny nets
DENSE_SHAPES = [
    ("outer", "LLGC", 3, 5, 37, 0.05, 0.2, True),          # (16, 32) instance, K not a multiple of 16
    ("outer", "LQGC", 17, 33, 16, 0.05, 0.05, True),       # N = 1; (32, 64)
    ("inner", "LLGC", 64, 64, 100, 0.05, 0.15, True),      # exact (64, 64)
    ("inner", "DoubleWell_multidim", 65, 30, 50, 0.05, 0.1, False),   # -> (128, 32), elementwise drift, non-adaptive image
    ("outer", "LLGC", 130, 40, 24, 0.05, 0.1, True),       # -> (256, 64): 16 state blocks, dense A and B
    ("inner", "LQGC", 200, 16, 20, 0.05, 0.1, True),       # -> (256, 32), running + terminal quadratic costs
    ("outer", "LLGC", 20, 30, 5000, 0.05, 0.15, True),     # many workgroups (313 tiles)
]


@pytest.mark.parametrize("mode,kind,d,H,K,dt,T,adaptive", DENSE_SHAPES)
def test_dense_shape_sweep_matches_oracle(mode, kind, d, H, K, dt, T, adaptive):
    if kind == "DoubleWell_multidim":
        kwargs = dict(d=d, d_1=d // 2, d_2=d - d // 2, T=T, eta=0.05, kappa=1.0)
    elif kind == "LQGC":
        kwargs = dict(d=d, off_diag=0.05, T=T, seed=42, delta_t=dt)
    else:
        kwargs = dict(d=d, off_diag=0.3 / d ** 0.5, T=T, seed=42)
    solver = dict(loss_method="log-variance", time_approx=mode, adaptive_forward_process=adaptive, detach_forward=True,
                  early_stopping_time=None, L=1, lr=0.002, seed=42, delta_t=dt, K=K, u_l2_error_flag=False)
    case = dict(name="dsweep", family="solver", problem=dict(kind=kind, kwargs=kwargs), solver=solver)
    if mode == "inner":
        case["net"] = dict(kind="densenet", arch=[H, H], seed=5)
    model = make_pkg_solver(case, dev(), backend="native", L=1)
    oprob, ocfg, omodels = make_oracle(case, L=1)
    if mode == "outer":                                   # arch is not a Solver keyword: swap the per-step nets on both sides
        model.z_n = [psp.DenseNet(d_in=d, d_out=d, lr=0.002, arch=[H, H], seed=5 + n).to(dev()) for n in range(model.N)]
        model.update_Phis()
        z = [orc.DenseNetOracle(d, d, 0.002, arch=[H, H], seed=5 + n) for n in range(model.N)]
        omodels = (z, omodels[1], omodels[2])
    model.train()
    assert model.plan_name == "native" and isinstance(model._native_plan, psp.plan_dense_native.DenseNativePlan)
    plan = model._native_plan
    ref = orc.hjb_train(oprob, ocfg, step_models=omodels, trace=True)
    tr = ref["traces"][0]
    D = plan.D.cpu()
    assert D.shape == tr["D"].shape
    assert float((D - tr["D"]).abs().max()) <= 2e-5 * max(1.0, float(tr["D"].abs().max()))
    g, g_ref = plan.grad.cpu(), torch.cat([x.reshape(-1) for x in tr["grads"]])
    assert g.shape == g_ref.shape
    assert float((g - g_ref).abs().max()) <= 2e-4 * float(g_ref.abs().max()), (plan.d_pad, plan.H_pad)
    cond = float((tr["D"].double() ** 2).mean()) / max(abs(ref["loss_log"][0]), 1e-30)
    assert math.isclose(model.loss_log[0], ref["loss_log"][0], rel_tol=min(1e-4, max(2e-5, 4 * 6e-8 * cond)))


# (time_approx, problem kind, d, H, K, delta_t, T, loss): gradients through the state path on instances the hand-written
# backward covers (d <= 128): padded shapes, ragged K, the two-launch column split, elementwise and dense drifts, running costs
ATTACHED_SHAPES = [
    ("outer", "LLGC", 3, 5, 37, 0.05, 0.2, "log-variance"),
    ("outer", "LQGC", 17, 33, 16, 0.05, 0.05, "moment"),                  # N = 1; running + terminal quadratic costs
    ("inner", "LLGC", 64, 64, 100, 0.05, 0.15, "log-variance"),           # exact (64, 64)
    ("inner", "DoubleWell_multidim", 65, 30, 50, 0.05, 0.1, "log-variance"),   # -> (128, 32), drift Jacobian of the double well
    ("outer", "LLGC", 100, 30, 200, 0.02, 0.1, "relative_entropy"),       # -> (112, 32): the bench instance, nu weights
    ("inner", "LQGC", 100, 64, 40, 0.05, 0.1, "cross_entropy"),           # -> (112, 64): column-split backward, explicit wT
    ("outer", "LLGC", 20, 30, 5000, 0.05, 0.15, "log-variance"),          # many workgroups (313 tiles)
]


@pytest.mark.parametrize("mode,kind,d,H,K,dt,T,loss", ATTACHED_SHAPES)
def test_attached_shape_sweep_matches_oracle(mode, kind, d, H, K, dt, T, loss):
    if kind == "DoubleWell_multidim":
        kwargs = dict(d=d, d_1=d // 2, d_2=d - d // 2, T=T, eta=0.05, kappa=1.0)
    elif kind == "LQGC":
        kwargs = dict(d=d, off_diag=0.05, T=T, seed=42, delta_t=dt)
    else:
        kwargs = dict(d=d, off_diag=0.3 / d ** 0.5, T=T, seed=42)
    solver = dict(loss_method=loss, time_approx=mode, adaptive_forward_process=True, detach_forward=False,
                  early_stopping_time=None, L=1, lr=0.002, seed=42, delta_t=dt, K=K, u_l2_error_flag=False)
    case = dict(name="asweep", family="solver", problem=dict(kind=kind, kwargs=kwargs), solver=solver)
    if mode == "inner":
        case["net"] = dict(kind="densenet", arch=[H, H], seed=5)
    model = make_pkg_solver(case, dev(), backend="native", L=1)
    oprob, ocfg, omodels = make_oracle(case, L=1)
    if mode == "outer":
        model.z_n = [psp.DenseNet(d_in=d, d_out=d, lr=0.002, arch=[H, H], seed=5 + n).to(dev()) for n in range(model.N)]
        model.update_Phis()
        z = [orc.DenseNetOracle(d, d, 0.002, arch=[H, H], seed=5 + n) for n in range(model.N)]
        omodels = (z, omodels[1], omodels[2])
    model.train()
    assert model.plan_name == "native" and isinstance(model._native_plan, psp.plan_dense_native.DenseNativePlan)
    plan = model._native_plan
    assert plan.attached
    ref = orc.hjb_train(oprob, ocfg, step_models=omodels, trace=True)
    tr = ref["traces"][0]
    D, D_ref = plan.D.cpu(), (-tr["Zsum_g"] if loss == "relative_entropy" else tr["D"])
    assert float((D - D_ref).abs().max()) <= 2e-5 * max(1.0, float(D_ref.abs().max()))
    g, g_ref = plan.grad.cpu(), torch.cat([x.reshape(-1) for x in tr["grads"]])
    assert g.shape == g_ref.shape
    err = float((g - g_ref).abs().max()) / float(g_ref.abs().max())
    print("attached %s %s d=%d H=%d K=%d %s: gradient rel err %.1e" % (mode, kind, d, H, K, loss, err))
    assert err <= 2e-4, (plan.d_pad, plan.H_pad, err)
    cond = float((D_ref.double() ** 2).mean()) / max(abs(ref["loss_log"][0]), 1e-30)
    assert math.isclose(model.loss_log[0], ref["loss_log"][0], rel_tol=min(1e-4, max(2e-5, 4 * 6e-8 * cond)))


def test_attached_outer_philox_is_deterministic_and_finite():
    """K = 2^14, d = 40, N = 20, device noise, the reference's default flags: two runs agree bitwise, the gradient is finite."""
    prob = psp.LLGC(d=40, off_diag=0.05, T=0.2, seed=42, device=dev())

    def make():
        return psp.Solver(name="big", problem=prob, loss_method="log-variance", time_approx="outer", L=2, lr=1e-3, seed=42,
                          delta_t=0.01, K=1 << 14, adaptive_forward_process=True, detach_forward=False, u_l2_error_flag=False,
                          verbose=False, device=dev(), backend="native", noise="philox")

    a, b = make(), make()
    a.train()
    b.train()
    assert a.plan_name == "native" and a._native_plan.attached
    assert a.loss_log == b.loss_log and all(math.isfinite(v) for v in a.loss_log)
    assert torch.equal(a._native_plan.grad, b._native_plan.grad) and bool(torch.isfinite(a._native_plan.grad).all())


@pytest.mark.parametrize("name", ["llgc_d12_outer_moment", "lqgc_d6_densenet_variance", "lqgc_d2_outer",
                                  "llgc_d100_densenet64_logvar"])          # the last one runs the two-launch column split
def test_kernel_backward_agrees_with_the_gemm_formulation(name, monkeypatch):
    """hjbd_bwd_kernel (hand-written: adjoint panels + weight-gradient outer products per (step, slice)) against the
    library-GEMM formulation of the same gradient (PSP_DENSE_BWD=gemm), same rollout, same noise."""
    case = load_golden(name)["case"]
    grads = {}
    for mode in ("kernel", "gemm"):
        monkeypatch.setenv("PSP_DENSE_BWD", mode)
        model = make_pkg_solver(case, dev(), backend="native", L=1)
        model.train()
        plan = model._native_plan
        assert plan.kernel_bwd == (mode == "kernel")
        grads[mode] = plan.grad.double().cpu()
    gk, gg = grads["kernel"], grads["gemm"]
    assert float((gk - gg).abs().max()) <= 2e-5 * float(gg.abs().max())


def test_outer_default_nets_at_d100_match_oracle():
    """The reference's constructor defaults at the benchmark dimension: time_approx='outer', 100 DenseNet(100 -> 100, [30, 30])
    nets (instance (112, 32), kernel backward), K = 512, N = 100, three full iterations on the reference noise stream."""
    case = dict(name="outer100", family="solver", problem=dict(kind="LLGC", kwargs=dict(d=100, off_diag=0.01, T=1.0, seed=42)),
                solver=dict(loss_method="log-variance", time_approx="outer", adaptive_forward_process=True, detach_forward=True,
                            early_stopping_time=None, L=3, lr=1e-3, seed=42, delta_t=0.01, K=512, u_l2_error_flag=False))
    model = make_pkg_solver(case, dev(), backend="native")
    oprob, ocfg, omodels = make_oracle(case)
    model.train()
    assert model.plan_name == "native" and model._native_plan.kernel_bwd
    ref = orc.hjb_train(oprob, ocfg, step_models=omodels)
    for got, want in zip(model.loss_log, ref["loss_log"]):
        assert math.isclose(got, want, rel_tol=1e-4), (model.loss_log, ref["loss_log"])


def test_unsupported_matrix_mode_falls_back_under_auto():
    """mlp_dtype='bf16' exists for MySequential controls only; with a DenseNet control the plan constructor raises
    PlanUnsupported -- backend='auto' then runs the composite torch plan like every other out-of-catalogue combination (never an
    error), backend='native' raises."""
    import warnings
    from path_space_pde_solver_amd.plan_native import PlanUnsupported
    rec = load_golden("lqgc_d2_outer")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model = make_pkg_solver(rec["case"], torch.device("cuda:0"), backend="auto", mlp_dtype="bf16")
        model.train()
    assert model.plan_name == "torch" and "bf16" in model.plan_reason
    assert any("composite torch plan" in str(x.message) for x in w)
    for got, want in zip(model.loss_log, rec["expected"]["loss_log"]):
        assert math.isclose(got, want, rel_tol=1e-4)
    with pytest.raises(PlanUnsupported):
        make_pkg_solver(rec["case"], torch.device("cuda:0"), backend="native", mlp_dtype="bf16").train()
