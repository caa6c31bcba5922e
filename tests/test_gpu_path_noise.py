"""store_path 4 (include/psp.h): the forward keeps X_n, h1, h2 only and the backward producers regenerate the Brownian increments
from the Philox counters.  The regenerated values ARE the stored ones, so everything downstream must be bit-identical to
store_path 1 -- D, loss, gradient, the parameters after Adam steps -- in both matrix modes, on a ragged K, under K-chunking and
when the range guard sends an iteration to the fp32-MFMA kernels (whose backward then regenerates as well).
Reference lines: solver.py:381 (the increments), :468-472 (what carries a gradient)."""
import pytest
import torch

from util_cases import psp

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def run(path_noise, mlp, K=16384 + 48, d=20, L=3, x0_scale=None, **kw):
    pb = psp.LLGC(d=d, off_diag=0.1, T=0.2, seed=42, device=dev())
    if x0_scale is not None:
        pb.X_0 = x0_scale * torch.ones(d)
    model = psp.Solver("pn", pb, lr=1e-3, L=L, K=K, delta_t=0.01, loss_method="log-variance", time_approx="inner",
                       adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42,
                       device=dev(), backend="native", noise="philox", widths=(30, 30), mlp_dtype=mlp, path_noise=path_noise, **kw)
    model.train()
    plan = model._native_plan
    return model, plan


@pytest.mark.parametrize("mlp", ["fp32", "f16x3", "bf16"])
def test_regenerated_increments_give_the_stored_ones_result(mlp):
    a, pa = run("store", mlp)
    b, pb = run("auto", mlp)
    assert not pa.regen_xi and pa.cfg.store_path == 1
    assert pb.regen_xi and pb.cfg.store_path == 4 and pb.matrix_mode == mlp
    assert a.loss_log == b.loss_log
    assert torch.equal(pa.D, pb.D) and torch.equal(pa.grad, pb.grad)
    assert torch.equal(pa.flat, pb.flat)                                   # parameters after three Adam steps
    assert float(pa.grad.abs().max()) > 0


def test_regeneration_under_k_chunking():
    a, pa = run("store", "f16x3", K=40000, path_chunks=3, chunk_mode="two_gradient")
    b, pb = run("auto", "f16x3", K=40000, path_chunks=3, chunk_mode="two_gradient")
    assert pa.n_chunks == 3 and pb.n_chunks == 3 and pb.regen_xi
    assert a.loss_log == b.loss_log and torch.equal(pa.grad, pb.grad) and torch.equal(pa.flat, pb.flat)
    c, pc = run("auto", "f16x3", K=40000, path_chunks=3, chunk_mode="recompute")
    assert pc.regen_xi and c.loss_log == a.loss_log
    assert float((pc.grad - pa.grad).abs().max()) <= 2e-5 * float(pa.grad.abs().max())     # (another summation order over chunks)


def test_regeneration_on_the_fp32_twin_after_a_range_fallback():
    """A state beyond the f16 range: the guarded split kernels hand every iteration to the fp32-MFMA twins, whose forward skips
    the xi store too and whose backward regenerates it."""
    a, pa = run("store", "f16x3", x0_scale=7.0e4)
    b, pb = run("auto", "f16x3", x0_scale=7.0e4)
    assert pb.regen_xi and a.range_fallback_iterations == 3 and b.range_fallback_iterations == 3
    assert all(map(lambda v: v == v and abs(v) < float("inf"), b.loss_log))
    assert a.loss_log == b.loss_log and torch.equal(pa.grad, pb.grad)
    f, pf = run("store", "fp32", x0_scale=7.0e4)
    assert f.loss_log == b.loss_log and torch.equal(pf.grad, pb.grad)


def test_where_the_increments_stay_in_the_store():
    """Small K (hipGraph regime), supplied noise, attached or non-adaptive runs and the wide family keep store_path 1 / 2."""
    m, p = run("auto", "fp32", K=1024)
    assert not p.regen_xi and p.cfg.store_path == 1
    pbm = psp.LLGC(d=20, off_diag=0.1, T=0.2, seed=42, device=dev())
    for kw in (dict(noise="reference"), dict(detach_forward=False), dict(adaptive_forward_process=False)):
        args = dict(lr=1e-3, L=1, K=16384 + 48, delta_t=0.01, loss_method="log-variance", time_approx="inner",
                    adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42, device=dev(),
                    backend="native", noise="philox", widths=(30, 30))
        args.update(kw)
        model = psp.Solver("pn", pbm, **args)
        model.train()
        assert not model._native_plan.regen_xi and model._native_plan.cfg.store_path in (1, 2), kw


def test_store_path_4_is_refused_where_the_image_is_not_the_increment():
    """The C ABI's own checks (include/psp.h): wide family, supplied noise, non-adaptive process, and the fused
    backward + Adam entry point (which takes no Philox seed)."""
    import ctypes as C
    nat = psp.native
    m, p = run("auto", "fp32", L=1)
    assert p.cfg.store_path == 4
    for field, value, code in (("noise_mode", nat.NOISE_SUPPLIED, -1), ("adaptive", 0, -1)):
        cfg = nat.HjbConfig.from_buffer_copy(p.cfg)
        setattr(cfg, field, value)
        rc, _, msg = nat.query_rc(cfg)
        assert rc == code and "store_path 4" in msg, (field, rc, msg)
    wide = psp.Solver("w", psp.LLGC(d=200, off_diag=0.01, T=0.1, seed=42, device=dev()), lr=1e-3, L=1, K=16384 + 16, delta_t=0.01,
                      loss_method="log-variance", time_approx="inner", adaptive_forward_process=True, detach_forward=True,
                      u_l2_error_flag=False, verbose=False, seed=42, device=dev(), backend="native", noise="philox", widths=(64, 64))
    wide.train()
    wp = wide._native_plan
    assert wp.family == 2 and not wp.regen_xi and wp.cfg.store_path == 1
    cfg = nat.HjbConfig.from_buffer_copy(wp.cfg)
    cfg.store_path = 4
    rc, _, msg = nat.query_rc(cfg)
    assert rc == -2 and "narrow kernel family" in msg, (rc, msg)
    # psp_hjb_rollout_bwd_step: arguments are checked before anything is launched
    z = torch.zeros(8, device=dev())
    rc = nat.load().psp_hjb_rollout_bwd_step(C.byref(p.cfg), nat.ptr(p.flat), nat.ptr(p.path), nat.ptr(p.D), nat.ptr(p.sums),
                                             nat.ptr(p.grad_partial), nat.ptr(p.grad), nat.ptr(p.m), nat.ptr(p.v), nat.ptr(z),
                                             nat.ptr(z), 1e-3, 0.9, 0.999, 1e-8, None)
    assert rc == -1 and "store_path 4" in nat.load().psp_last_error().decode()


def test_headline_shape_regenerated_equals_stored():
    """BASELINE.json's headline shape (d=100, H=64, K=65536, N=100, split products): two iterations with the increments kept in the
    path store and with them regenerated by the backward agree bit for bit in D, loss, gradient and parameters."""
    pb = psp.LLGC(d=100, off_diag=0.01, T=1.0, seed=42, device=dev())

    def go(path_noise):
        m = psp.Solver("hl", pb, lr=1e-3, L=2, K=65536, delta_t=0.01, loss_method="log-variance", time_approx="inner",
                       adaptive_forward_process=True, detach_forward=True, u_l2_error_flag=False, verbose=False, seed=42, device=dev(),
                       backend="native", noise="philox", widths=(64, 64), path_noise=path_noise)
        m.train()
        return m, m._native_plan

    a, pa = go("store")
    b, pb2 = go("auto")
    assert pa.cfg.store_path == 1 and pb2.cfg.store_path == 4 and pb2.matrix_mode == "f16x3"
    assert a.loss_log == b.loss_log and all(v == v for v in a.loss_log)
    assert torch.equal(pa.D, pb2.D) and torch.equal(pa.grad, pb2.grad) and torch.equal(pa.flat, pb2.flat)
    assert b.range_fallback_iterations == 0
