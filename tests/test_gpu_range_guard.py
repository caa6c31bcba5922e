"""Range safety of the split-product ('f16x3') kernels, the default matrix mode of the native plans.

The reference computes in fp32 end to end (solver.py:39-40).  An f16x3 operand beyond 65504 cannot be represented (hi = inf,
lo = -inf: every product it enters ends as inf - inf), so D_k of that trajectory is NaN.  With the range guard (include/psp.h:
range_flag; Solver(range_guard=True), the default) the library raises a device flag from the non-finite partial sums and the
fp32-MFMA kernels, enqueued behind the split ones and predicated on that flag, redo the iteration:
  * a guarded run whose state leaves the f16 range equals the mlp_dtype='fp32' run BIT FOR BIT (same kernels did the work) and
    the CPU oracle to the usual bounds; the unguarded split kernels return NaN on the same input (so the test means something);
  * a guarded run that stays in range never takes the fallback and equals the unguarded run bit for bit;
  * hjb_bwd3_kernel takes its power-of-two gradient scale from a scan of ALL trajectory weights: a workgroup whose first round
    carries zero weights, and a weight 1e5 x the others, give the fp32 kernel's gradient (and the oracle's).
"""
import ctypes as C
import math

import pytest
import torch

from conftest import load_golden
from util_cases import make_oracle, make_pkg_solver, orc, psp

pytestmark = pytest.mark.gpu
nat = psp.native
BIG = 7.0e4                      # beyond the largest finite f16 (65504), far inside fp32


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def tile_per_wave_forward(monkeypatch):
    """The guard falls back to the tile-per-wave fp32 forward (hjb_fwd_kernel); the golden cases are small enough for the
    library to pick its small-K forwards in plain fp32 mode (equal to summation order only).  Pin the fp32 runs to the same
    kernel so that 'equal' below can mean bit-identical."""
    monkeypatch.setenv("PSP_FWD_VARIANT", "1")


def _hjb(name, mlp, big=True, **over):
    rec = load_golden(name)
    case = rec["case"]
    model = make_pkg_solver(case, dev(), backend="native", mlp_dtype=mlp, **over)
    if big:
        x0 = model.X_0.clone()
        x0[min(3, model.d - 1)] = BIG          # one state component outside the f16 range from step 0 on
        model.X_0 = x0
    model.train()
    assert model.plan_name == "native"
    return case, model


@pytest.mark.parametrize("name,over", [
    ("llgc_d100_h64_logvar", dict(loss_method="moment")),                      # narrow family: dense drift, dense sigma
    ("llgc_d20_diag_logvar", dict(loss_method="moment")),                      # diagonal drift: only the W1 product sees X
    ("llgc_d200_h64_logvar", dict(loss_method="moment")),                      # wide family
    ("llgc_d500_h64_logvar", dict(loss_method="moment")),                      # wide family, d > 256: hjbw_bwd_x3_kernel
    ("llgc_d100_h64_attached_logvar", dict(loss_method="moment")),             # adjoint sweep pair
    ("llgc_d100_h64_logvar", dict(loss_method="moment", path_chunks=3)),       # K-chunked, one flag per chunk launch
])
def test_state_beyond_the_f16_range_runs_on_the_fp32_kernels(name, over):
    L = 2
    case, ref = _hjb(name, "fp32", L=L, **over)
    assert all(math.isfinite(v) for v in ref.loss_log), ref.loss_log
    case, got = _hjb(name, "f16x3", L=L, **over)
    assert got._native_plan.matrix_mode == "f16x3" and got._native_plan.range_flag is not None
    assert got.loss_log == ref.loss_log, (got.loss_log, ref.loss_log)
    n_launch = L * max(1, got._native_plan.n_chunks) * (2 if got._native_plan.chunk_mode == "recompute" else 1)
    assert got.range_fallback_iterations == n_launch, (got.range_fallback_iterations, n_launch)
    assert torch.equal(got._native_plan.grad, ref._native_plan.grad)
    for p, q in zip(got.z_n.parameters(), ref.z_n.parameters()):
        assert torch.equal(p, q)
    # the unguarded split kernels cannot represent this state
    case, raw = _hjb(name, "f16x3", L=1, range_guard=False, **over)
    assert raw._native_plan.range_flag is None and not math.isfinite(raw.loss_log[0])


def test_guarded_overflow_run_matches_the_oracle():
    """... and the fp32 result it falls back to is the reference's: moment loss (the log-variance of a state near 7e4 is
    cancellation noise in the reference's own fp32 mean(D^2) - mean(D)^2)."""
    rec = load_golden("llgc_d100_h64_logvar")
    case = rec["case"]
    case, got = _hjb("llgc_d100_h64_logvar", "f16x3", L=2, loss_method="moment")
    oprob, ocfg, omodels = make_oracle(case, L=2)
    ocfg.loss_method = "moment"
    x0 = oprob.X_0.clone()
    x0[3] = BIG
    oprob.X_0 = x0
    ref = orc.hjb_train(oprob, ocfg, step_models=omodels, trace=True)
    for l in range(2):
        assert math.isclose(got.loss_log[l], ref["loss_log"][l], rel_tol=1e-4), (got.loss_log, ref["loss_log"])
    assert got.range_fallback_iterations == 2
    assert float(ref["traces"][-1]["X"][0].abs().max()) >= BIG          # the state really was out there


def test_auto_mode_is_guarded_at_large_K():
    """'auto' picks the split kernels once there are more than two tiles per CU: same guard."""
    K = 16 * (2 * torch.cuda.get_device_properties(dev()).multi_processor_count + 16)
    case, ref = _hjb("llgc_d100_h64_logvar", "fp32", L=1, K=K, noise="philox", loss_method="moment")
    case, got = _hjb("llgc_d100_h64_logvar", "auto", L=1, K=K, noise="philox", loss_method="moment")
    assert got._native_plan.matrix_mode == "f16x3" and got.range_fallback_iterations == 1
    assert math.isfinite(ref.loss_log[0]) and got.loss_log == ref.loss_log


@pytest.mark.parametrize("name", ["llgc_d100_h64_logvar", "llgc_d200_h64_logvar", "llgc_d100_h64_attached_logvar"])
def test_in_range_run_never_takes_the_fallback(name):
    case, raw = _hjb(name, "f16x3", big=False, range_guard=False)
    case, got = _hjb(name, "f16x3", big=False)
    assert got._native_plan.range_flag is not None and got.range_fallback_iterations == 0
    assert got.loss_log == raw.loss_log
    assert torch.equal(got._native_plan.grad, raw._native_plan.grad)
    exp = load_golden(name)["expected"]["loss_log"]
    for a, b in zip(got.loss_log, exp):
        assert math.isclose(a, b, rel_tol=1e-4)


def test_dense_control_rollout_is_guarded():
    """time_approx='outer' (hjbd_fwd_kernel<.., X3>): nets scaled down so that the relu^2 layers stay finite in fp32 at |x| = 7e4."""
    rec = load_golden("llgc_d12_outer_moment")

    def run(mlp, **kw):
        model = make_pkg_solver(rec["case"], dev(), backend="native", mlp_dtype=mlp, L=2, **kw)
        with torch.no_grad():
            for net in model.z_n:
                for p in net.parameters():
                    p.mul_(1e-2)
        x0 = model.X_0.clone()
        x0[3] = BIG
        model.X_0 = x0
        model.train()
        assert model.plan_name == "native"
        return model

    ref, got = run("fp32"), run("f16x3")
    assert all(math.isfinite(v) for v in ref.loss_log), ref.loss_log
    assert got._native_plan.matrix_mode == "f16x3" and got.range_fallback_iterations == 2
    assert got.loss_log == ref.loss_log
    raw = run("f16x3", range_guard=False)
    assert not math.isfinite(raw.loss_log[0])


def test_general_solver_is_guarded():
    """GeneralSolver (gen_fwd_kernel / gen_bwd2_kernel<.., X3>): initial points on a ball of radius 3e5, value net scaled down."""
    rec = load_golden("heat_d6_diffusion")
    case = rec["case"]

    def run(mlp, **kw):
        prob = getattr(psp, case["problem"]["kind"])(device=dev(), **case["problem"]["kwargs"])
        prob.boundary_distance = 3.0e5
        s = dict(case["solver"])
        s.update(L=2, mlp_dtype=mlp, **kw)
        model = psp.GeneralSolver(problem=prob, name=case["name"], verbose=False, device=dev(), backend="native", **s)
        if "net" in case:
            model.V = psp.DenseNet(d_in=prob.d + 1, d_out=1, lr=s["lr"], arch=case["net"]["arch"], seed=case["net"]["seed"]).to(dev())
        with torch.no_grad():
            for p in model.V.parameters():
                p.mul_(1e-1)
        model.train()
        assert model.plan_name == "native"
        return model

    ref, got = run("fp32"), run("f16x3")
    assert all(math.isfinite(v) for v in ref.loss_log), ref.loss_log
    assert got._gen_plan.matrix_mode == "f16x3" and got.range_fallback_iterations == 2
    assert got.loss_log == ref.loss_log and got.K_log == ref.K_log
    assert torch.equal(got._gen_plan.grad, ref._gen_plan.grad)
    raw = run("f16x3", range_guard=False)
    assert not math.isfinite(raw.loss_log[0])
    ok = run("f16x3")                                      # (and a second guarded model starts with a fresh counter)
    assert ok.range_fallback_iterations == 2


# ---- gradient scale of hjb_bwd3_kernel ---------------------------------------------------------------------------------------
def _bwd(plan, model, params, w, mlp):
    """psp_hjb_rollout_bwd on the plan's own path store with caller-supplied trajectory weights (PSP_LOSS_WEIGHTS) and the
    parameters the store was written with."""
    cfg = nat.HjbConfig.from_buffer_copy(plan.cfg)
    cfg.loss_kind, cfg.mlp_dtype, cfg.range_flag = nat.LOSS_WEIGHTS, mlp, None
    sizes = nat.query(cfg)
    grad = torch.zeros(plan.pad.Pp, dtype=torch.float32, device=dev())
    part = torch.zeros(sizes.grad_partial_bytes // 4, dtype=torch.float32, device=dev())
    nat.check(nat.load().psp_hjb_rollout_bwd(C.byref(cfg), nat.ptr(params), None, int(model.seed), 0, nat.ptr(plan.path),
                                              nat.ptr(w), nat.ptr(plan.sums), nat.ptr(part), nat.ptr(grad),
                                              nat.stream_ptr(dev())), "psp_hjb_rollout_bwd")
    torch.cuda.synchronize()
    rows = part[:sizes.bwd_workgroups * plan.pad.Pp].view(sizes.bwd_workgroups, plan.pad.Pp)
    return grad, rows


def _oracle_weighted_gradient(case, K, model, w):
    """Gradient of sum_k w_k D_k by the oracle's autograd, on the Philox stream the kernels used."""
    N, d = model.N, model.d
    xi = torch.empty(N + 1, K, d, device=dev())
    nat.check(nat.load().psp_philox_normal_fill(nat.ptr(xi), N, K, d, 0, int(model.seed), 0, None), "fill")
    torch.cuda.synchronize()
    oprob, ocfg, omodels = make_oracle(case, L=1)
    ocfg.K = K
    keep = orc.hjb_loss
    wc = w.cpu()
    orc.hjb_loss = lambda kind, D, Y, gX, **kw: (wc * D).sum()
    try:
        ref = orc.hjb_train(oprob, ocfg, step_models=omodels, noise=[xi.cpu().permute(1, 2, 0).contiguous()], trace=True)
    finally:
        orc.hjb_loss = keep
    return torch.cat([g.reshape(-1) for g in ref["traces"][0]["grads"]])


@pytest.mark.parametrize("kind", ["zero_first_round", "outlier_1e5", "plain"])
def test_bwd3_gradient_scale_comes_from_all_weights(kind):
    case = load_golden("llgc_d100_h64_logvar")["case"]
    K = 1040                               # 65 tiles x 50 steps = 813 rounds on 256 workgroups, tiles rotate through them
    model = make_pkg_solver(case, dev(), backend="native", noise="philox", L=1, K=K, mlp_dtype="f16x3")
    from util_cases import flat_params
    params0 = flat_params(model.z_n).to(dev())           # train() below takes one Adam step; the path store is of THESE
    model.train()
    plan = model._native_plan
    assert plan.matrix_mode == "f16x3" and plan.pad.identity
    g = torch.Generator(device="cpu").manual_seed(7)
    w = (torch.randn(K, generator=g) * (2.0 / K)).to(dev())
    if kind == "zero_first_round":
        w[:64] = 0.0                       # workgroup 0's first round = tiles 0..3 of step 0: all weights zero
    elif kind == "outlier_1e5":
        w[777] = 1.0e5 * (2.0 / K)
    g3, rows3 = _bwd(plan, model, params0, w, nat.MLP_F16X3)
    g2, rows2 = _bwd(plan, model, params0, w, nat.MLP_FP32)
    assert torch.isfinite(g3).all()
    scale = float(g2.abs().max())
    assert float((g3 - g2).abs().max()) <= 2e-5 * scale, float((g3 - g2).abs().max()) / scale
    # per workgroup (same round assignment in both kernels): this is where a first-round scale lost its bits
    rs = rows2.abs().max(dim=1).values.clamp_min(1e-30)
    rel = ((rows3 - rows2).abs().max(dim=1).values / rs)
    assert float(rel.max()) <= 1e-4, (int(rel.argmax()), float(rel.max()))
    g_ref = _oracle_weighted_gradient(case, K, model, w).to(dev())
    assert float((g3 - g_ref).abs().max()) <= 2e-4 * float(g_ref.abs().max())


@pytest.mark.parametrize("kind", ["zero_first_round", "outlier_1e5"])
def test_wide_split_backward_gradient_scale_comes_from_all_weights(kind):
    """hjbw_bwd2x_kernel (wide family, d <= 256) scales the trajectory weights by one power of two per launch, taken from a scan of
    ALL weights: a workgroup whose first round carries zero weights, and one weight 1e5 x the others, against the fp32-MFMA
    kernel per workgroup and against the oracle's gradient of sum_k w_k D_k."""
    case = load_golden("llgc_d200_h64_logvar")["case"]
    K = 1040
    model = make_pkg_solver(case, dev(), backend="native", noise="philox", L=1, K=K, mlp_dtype="f16x3")
    from util_cases import flat_params
    params0 = flat_params(model.z_n).to(dev())
    model.train()
    plan = model._native_plan
    assert plan.matrix_mode == "f16x3" and plan.family == 2 and plan.pad.identity
    g = torch.Generator(device="cpu").manual_seed(7)
    w = (torch.randn(K, generator=g) * (2.0 / K)).to(dev())
    if kind == "zero_first_round":
        w[:64] = 0.0
    else:
        w[777] = 1.0e5 * (2.0 / K)
    g3, rows3 = _bwd(plan, model, params0, w, nat.MLP_F16X3)
    g2, rows2 = _bwd(plan, model, params0, w, nat.MLP_FP32)
    assert torch.isfinite(g3).all() and not torch.equal(g3, g2)
    scale = float(g2.abs().max())
    assert float((g3 - g2).abs().max()) <= 2e-5 * scale, float((g3 - g2).abs().max()) / scale
    rs = rows2.abs().max(dim=1).values.clamp_min(1e-30)
    rel = ((rows3 - rows2).abs().max(dim=1).values / rs)
    assert float(rel.max()) <= 1e-4, (int(rel.argmax()), float(rel.max()))
    g_ref = _oracle_weighted_gradient(case, K, model, w).to(dev())
    assert float((g3 - g_ref).abs().max()) <= 2e-4 * float(g_ref.abs().max())


# ---- backward side of the guard (round 4): the forward stays finite, the split-product BACKWARD does not ------------
def test_backward_overflow_with_a_finite_forward_is_caught():
    """relu^2 pre-activations around 200: h = r^2 = 4e4 is a finite f16 forward operand, so the forward raises no flag -- but the
    adjoint dz2 = (W3 G) 2 r of hjbd_bwd_kernel<.., X3> (G scaled to a top magnitude in [64, 128), |W3| ~ 1) leaves the f16
    range.  Every split-product backward kernel checks what it writes into its partial gradient and raises the same flag; the
    fp32-MFMA twin behind it redoes the pass BEFORE the optimiser step: finite gradient (that of the fp32 kernels on the same
    path store), finite parameters, one fallback counted.  Unguarded, the same run leaves NaN parameters behind."""
    rec = load_golden("llgc_d12_outer_moment")

    def run(mlp, **kw):
        # one time step (dt = T): the state stays at X_0 = 0 inside the net, only the biases set the pre-activations
        model = make_pkg_solver(rec["case"], dev(), backend="native", mlp_dtype=mlp, L=1, delta_t=0.2, **kw)
        assert model.N == 1
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for net in model.z_n:
                W1, b1, W2, b2, W3, b3 = list(net.W)
                b2.fill_(200.0)
                W3.copy_(torch.randn(W3.shape, generator=g).to(W3.device))
        model.train()
        assert model.plan_name == "native"
        return model

    ref, got = run("fp32"), run("f16x3")
    g_ref, g = ref._native_plan.grad, got._native_plan.grad
    assert math.isfinite(ref.loss_log[0]) and bool(torch.isfinite(g_ref).all())
    assert got._native_plan.matrix_mode == "f16x3"
    assert math.isfinite(got.loss_log[0]) and math.isclose(got.loss_log[0], ref.loss_log[0], rel_tol=1e-4)
    assert bool(torch.isfinite(g).all()), "non-finite gradient reached the optimiser"
    assert got.range_fallback_iterations == 1
    assert float((g - g_ref).abs().max()) <= 2e-4 * float(g_ref.abs().max())
    for net in got.z_n:
        assert all(bool(torch.isfinite(p).all()) for p in net.parameters())
    raw = run("f16x3", range_guard=False)                          # the test means something: unguarded, the backward overflows
    assert not (math.isfinite(raw.loss_log[0]) and bool(torch.isfinite(raw._native_plan.grad).all()))


def test_backward_overflow_general_solver():
    """The same window for gen_bwd2_kernel<.., X3>: second-layer bias at 200, an O(1) output layer; one rollout step."""
    rec = load_golden("heat_d6_diffusion")
    case = rec["case"]

    def run(mlp, **kw):
        prob = getattr(psp, case["problem"]["kind"])(device=dev(), **case["problem"]["kwargs"])
        s = dict(case["solver"])
        s.update(L=1, N=1, mlp_dtype=mlp, **kw)
        model = psp.GeneralSolver(problem=prob, name=case["name"], verbose=False, device=dev(), backend="native", **s)
        g = torch.Generator().manual_seed(6)
        with torch.no_grad():
            W1, b1, W2, b2, W3, b3 = list(model.V.W)
            b2.fill_(200.0)
            W3.copy_(torch.randn(W3.shape, generator=g).to(W3.device))
        model.train()
        assert model.plan_name == "native"
        return model

    ref, got = run("fp32"), run("f16x3")
    g_ref, g = ref._gen_plan.grad, got._gen_plan.grad
    assert math.isfinite(ref.loss_log[0]) and bool(torch.isfinite(g_ref).all())
    assert got._gen_plan.matrix_mode == "f16x3" and math.isfinite(got.loss_log[0])
    assert bool(torch.isfinite(g).all()), "non-finite gradient reached the optimiser"
    assert float((g - g_ref).abs().max()) <= 5e-4 * float(g_ref.abs().max())
    assert all(bool(torch.isfinite(p).all()) for p in got.V.parameters())
