"""hipGraph replay of the launch-bound iteration (plan_native.HjbNativePlan._iteration_graph; include/psp.h psp_iter_state)
against the eager launch sequence: same kernels, the per-iteration quantities (Philox iteration index, loss-log slot, Adam
step / bias corrections) read from device memory instead of host arguments.

Equality: D of every iteration and the loss log agree to 1e-6 relative (the rollout kernels are the same; Adam's bias
corrections come from running fp64 products beta^step instead of pow(), which can move a parameter by one ulp)."""
import math

import pytest
import torch

from conftest import load_golden
from util_cases import flat_params, make_pkg_solver

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", ["llgc_d100_h64_logvar", "lqgc_d2_moment", "lqgc_d33_h50_logvar",
                                  "llgc_d20_relative_entropy_detached", "dw_d10_logvar",
                                  # attached forward process (the reference's default flags): the adjoint sweep and its
                                  # trajectory weights are part of the captured iteration
                                  "llgc_d100_h64_attached_logvar", "dw_d10_attached_moment", "lqgc_d4_relative_entropy"])
def test_graph_replay_equals_eager_iterations(name):
    case = load_golden(name)["case"]
    L = 7
    runs = {}
    for use in (False, True):
        m = make_pkg_solver(case, dev(), backend="native", noise="philox", L=L, use_graph=use)
        m.train()
        plan = m._native_plan
        assert plan.graph_active == use
        runs[use] = (m.loss_log, flat_params(m.z_n), plan.D.clone(), list(m.Y_0_log), plan.step)
    (le, pe, De, ye, se), (lg, pg, Dg, yg, sg) = runs[False], runs[True]
    assert se == sg == L and len(le) == len(lg) == L
    for a, b in zip(le, lg):
        assert math.isclose(a, b, rel_tol=1e-6), (le, lg)
    assert float((pe - pg).abs().max()) <= 1e-6 * max(1.0, float(pe.abs().max()))
    assert float((De - Dg).abs().max()) <= 1e-5 * max(1.0, float(De.abs().max()))
    for a, b in zip(ye, yg):
        assert math.isclose(a, b, rel_tol=1e-6, abs_tol=1e-8)


def test_graph_is_the_default_for_small_K_and_follows_the_loss_buffer():
    """'auto': K = 1024 (64 tiles) replays a graph, K = 16384 (1024 tiles > 2 per CU) launches eagerly; a second train()
    call (new loss buffer, continued Adam state) re-captures and keeps training."""
    case = load_golden("llgc_d100_h64_logvar")["case"]
    small = make_pkg_solver(case, dev(), backend="native", noise="philox", L=4, K=1024)
    small.train()
    assert small._native_plan.graph_active
    first = list(small.loss_log)
    small.train()                                         # continues from the trained weights
    assert len(small.loss_log) == 8 and small._native_plan.step == 8
    assert small.loss_log[4:] != first
    big = make_pkg_solver(case, dev(), backend="native", noise="philox", L=2, K=16384)
    big.train()
    assert not big._native_plan.graph_active
    # reference noise needs a host upload per iteration: never captured
    ref = make_pkg_solver(case, dev(), backend="native", noise="reference", L=2, K=1024, use_graph=True)
    ref.train()
    assert not ref._native_plan.graph_active


def test_adam_state_is_handed_to_the_nets_own_optimiser_and_back():
    """The reference keeps Adam's state in phi.optim (function_space.py:185).  After a native train() the net's optimiser holds
    the plan's moments, so (a) a composite-plan continuation and (b) a rebuilt native plan both continue the SAME optimiser:
    native 3 + native 3 iterations (second train() after the plan was dropped) equals native 6 iterations."""
    case = load_golden("lqgc_d2_logvar_noul2")["case"]
    six = make_pkg_solver(case, dev(), backend="native", noise="philox", L=6, use_graph=False)
    six.train()
    a = make_pkg_solver(case, dev(), backend="native", noise="philox", L=3, use_graph=False)
    a.train()
    st = a.z_n.optim.state[a.z_n.linears[0].weight]
    assert int(float(st["step"])) == 3 and float(st["exp_avg_sq"].abs().max()) > 0.0
    a._native_plan = None                                  # force a new plan: it must pick the optimiser state up again
    plan = a._choose_plan()
    assert plan.step == 3
    losses = torch.zeros(6, device=dev())
    for l in range(3, 6):                                  # same Philox iteration indices as the 6-iteration run
        plan.iteration(l, losses)
    torch.cuda.synchronize()
    got = a.loss_log + losses[3:].cpu().tolist()
    for x, y in zip(got, six.loss_log):
        assert math.isclose(x, y, rel_tol=1e-6), (got, six.loss_log)
    assert float((flat_params(a.z_n) - flat_params(six.z_n)).abs().max()) <= 1e-6
