"""Corners of the padded kernel-instance grid (native_shapes.py) against the CPU oracle: hidden widths that give 1, 2, 3
and 4 hidden blocks, several state-block counts, ragged trajectory counts, both the feature-split forward (few tiles) and
the tile-per-wave forward, detached and attached forward process.  First-iteration D and gradient, same tolerances as
test_gpu_parity.py."""
import math

import pytest
import torch

from util_cases import flat_params, make_oracle, make_pkg_solver, orc

pytestmark = pytest.mark.gpu
HJB = dict(loss_method="log-variance", time_approx="inner", adaptive_forward_process=True, detach_forward=True,
           early_stopping_time=None)

# (problem kind, d, widths, K, detach_forward)
SHAPES = [
    ("LLGC", 5, [10, 10], 40, True),        # -> (8, 30) or (16, 16): 1 hidden block after padding to 16
    ("LLGC", 12, [16, 16], 50, True),       # exact H = 16: one hidden block
    ("LQGC", 20, [40, 40], 72, True),       # -> (32, 48): three hidden blocks
    ("LLGC", 60, [48, 48], 100, False),     # -> (64, 48), attached (adjoint sweep) with three hidden blocks
    ("LQGC", 90, [16, 16], 33, True),       # -> (96, 16): six state blocks, one hidden block
    ("DoubleWell_multidim", 30, [64, 64], 130, False),   # -> (32, 64), attached, elementwise drift
    ("LQGC", 20, [40, 40], 12000, True),    # many tiles: tile-per-wave forward, several backward rounds per workgroup
    ("LLGC", 40, [48, 48], 9000, False),    # same with the adjoint sweep
    ("LLGC", 112, [20, 20], 48, True),
    ("LLGC", 200, [64, 64], 80, False),     # wide family with the adjoint sweep (hjbw_adj_kernel)
    ("LQGC", 130, [40, 40], 40, False),     # -> wide (192, 64), running cost + quadratic terminal cost, attached
    # hjbw_bwd2_kernel (role-specialised wide backward, d <= 256): several rounds per workgroup, so that the consumers' register
    # ring runs through block AND round boundaries, on every instance it is built for
    ("LLGC", 120, [64, 64], 9000, True),    # -> wide (128, 64)
    ("LQGC", 180, [64, 64], 8200, True),    # -> wide (192, 64), ragged last tile
    ("LLGC", 200, [64, 64], 9000, True),    # (200, 64)
    ("LLGC", 250, [50, 50], 9000, True),    # -> wide (256, 64)
    ("LLGC", 500, [64, 64], 36, False),      # (112, 32) does not fit the LDS with dense A and B -> wide family (128, 64)
]


@pytest.mark.parametrize("kind,d,widths,K,detach", SHAPES)
def test_padded_shapes_match_oracle(kind, d, widths, K, detach):
    if kind == "DoubleWell_multidim":
        kwargs = dict(d=d, d_1=d // 2, d_2=d - d // 2, T=0.2, eta=0.05, kappa=1.0)
    elif kind == "LQGC":
        kwargs = dict(d=d, off_diag=0.05, T=0.2, seed=42, delta_t=0.05)
    else:
        kwargs = dict(d=d, off_diag=0.3 / d ** 0.5, T=0.2, seed=42)
    case = dict(name="sweep", family="solver", problem=dict(kind=kind, kwargs=kwargs),
                solver=dict(HJB, detach_forward=detach, L=1, lr=0.002, seed=42, delta_t=0.05, K=K, u_l2_error_flag=False),
                net=dict(kind="tanh_mlp", widths=widths, seed=123))
    model = make_pkg_solver(case, torch.device("cuda:0"), backend="native", L=1)
    oprob, ocfg, omodels = make_oracle(case, L=1)
    assert torch.equal(flat_params(model.z_n), flat_params(omodels[0]))
    model.train()
    assert model.plan_name == "native"
    plan = model._native_plan
    ref = orc.hjb_train(oprob, ocfg, step_models=omodels, trace=True)
    tr = ref["traces"][0]
    D = plan.D.cpu()
    scale = max(1.0, float(tr["D"].abs().max()))
    assert float((D - tr["D"]).abs().max()) <= 2e-5 * scale
    g = plan.grad.cpu()
    g_ref = torch.cat([x.reshape(-1) for x in tr["grads"]])
    assert g.shape == g_ref.shape
    assert float((g - g_ref).abs().max()) <= 2e-4 * float(g_ref.abs().max()), (plan.d_pad, plan.H_pad, plan.family)
    assert math.isclose(model.loss_log[0], ref["loss_log"][0], rel_tol=1e-4)
