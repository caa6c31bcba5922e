"""GeneralSolver kernels beyond the golden cases:
  * MANY rounds per persistent workgroup (the golden cases are one round each, which cannot see cross-round pipeline bugs):
    K = 32768 / 20000 trajectories, N = 20, against the CPU oracle's autograd on the reference's noise stream;
  * the bf16 MFMA modes (BASELINE.json configs[2]) against the fp32 kernels and the goldens, with their own tolerance.
(The two-workgroups-per-CU backward gen_bwd_kernel that this file used to cross-check is no longer part of the shipped
library: -DPSP_LEGACY_BWD diagnostic builds only.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("K,H,loss", [(32768, 24, 'diffusion'), (20000, 32, 'BSDE')])
def test_many_rounds_match_the_oracle(K, H, loss):
    import math

    import torch

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util_cases import orc, psp
    dev = torch.device('cuda:0')
    pk = dict(d=10, d_1=5, d_2=5, T=0.3, eta=1.0, kappa=1.0, modus='HJB')
    prob = psp.DoubleWell_multidim_for_general_solver(device=dev, **pk)
    m = psp.GeneralSolver(prob, 'rounds', seed=42, delta_t=0.01, N=20, lr=1e-3, L=1, K=K, K_boundary=50,
                          loss_method=loss, verbose=False, device=dev, backend='native', noise='reference')
    m.V = psp.DenseNet(d_in=11, d_out=1, lr=1e-3, arch=[H, H], seed=42).to(dev)
    m.train()
    plan = m._gen_plan
    nround = ((m.N + 1) * ((K + 15) // 16) + 3) // 4
    assert nround >= 4 * plan.sizes.bwd_workgroups, (nround, plan.sizes.bwd_workgroups)
    torch.set_num_threads(16)
    oprob = orc.make_problem("DoubleWell_multidim_for_general_solver", **pk)
    cfg = orc.GeneralConfig(K=K, N=20, delta_t=0.01, lr=1e-3, L=1, seed=42, K_boundary=50, alpha=(1.0, 1.0, 1.0),
                            loss_method=loss)
    ref = orc.general_train(oprob, cfg, V=orc.general_build(oprob, cfg, arch=[H, H]), trace=True)
    assert m.K_log == ref["K_log"]
    assert math.isclose(m.loss_log[0], ref["loss_log"][0], rel_tol=5e-5), (m.loss_log, ref["loss_log"])
    g_ref = torch.cat([g.reshape(-1) for g in ref["traces"][0]["grads"]])
    g = plan.grad.cpu()
    err = float((g - g_ref).abs().max()) / float(g_ref.abs().max())
    print("K=%d H=%d %s: %d rounds on %d workgroups, gradient rel err %.1e" % (K, H, loss, nround, plan.sizes.bwd_workgroups, err))
    assert err <= 5e-4


def test_bf16_forward_products_track_the_fp32_kernels():
    """BASELINE.json configs[2] names a bf16 MFMA MLP path: mlp_dtype='bf16' runs the value-net products of the forward
    rollout (V, grad_x V, tangent pass) on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; state, Y, the path store and the
    backward stay fp32.  Its OWN tolerance (bf16 operands carry 8 mantissa bits): loss within 2 % of the fp32 kernels and
    of the reference's golden run over the logged iterations, gradient direction cosine >= 0.999, active-step counts equal."""
    import math

    import torch

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden
    from test_gpu_general import build
    for name in ("dwgen_d10_diffusion", "allencahn_d20_default_diffusion", "dwgen_d40_h50_bsde", "heat_d6_diffusion",
                 "dwgen_d100_h64_diffusion"):           # gen_*<100, 64, bf16>: the instances BASELINE configs[2] names
        rec = load_golden(name)
        res = {}
        for dt in ("fp32", "bf16_fwd", "bf16"):
            prob, model = build(rec["case"], mlp_dtype=dt)
            model.train()
            assert model.plan_name == "native"
            res[dt] = (model.loss_log, model.K_log, model._gen_plan.grad.double().cpu())
        g32 = res["fp32"][2]
        for dt in ("bf16_fwd", "bf16"):                      # forward products only / forward + backward products
            g16 = res[dt][2]
            cos = float(torch.dot(g32, g16) / (g32.norm() * g16.norm()))
            assert res["fp32"][1] == res[dt][1], name
            assert cos >= 0.999, (name, dt, cos)
            for a, b, g in zip(res[dt][0], res["fp32"][0], rec["expected"]["loss_log"]):
                assert math.isclose(a, b, rel_tol=2e-2) and math.isclose(a, g, rel_tol=2e-2), (name, dt, res[dt][0], res["fp32"][0])
            assert res[dt][0] != res["fp32"][0], (name, dt)    # the bf16 path really ran
        assert not torch.equal(res["bf16"][2], res["bf16_fwd"][2]), name


def test_bf16_backward_many_rounds():
    """Every backward workgroup runs MANY rounds (the golden cases give one): the bf16 outer products pair the sample
    blocks of a round, so round boundaries, ragged tails (K not a multiple of 64) and the final-point blocks all occur."""
    import torch

    sys.path.insert(0, ROOT)
    import path_space_pde_solver_amd as psp
    dev = torch.device("cuda:0")
    prob = psp.DoubleWell_multidim_for_general_solver(d=10, d_1=5, d_2=5, T=0.3, eta=1.0, kappa=1.0, modus="HJB", device=dev)
    res = {}
    for dt in ("fp32", "bf16"):
        m = psp.GeneralSolver(prob, "var", seed=42, delta_t=0.01, N=20, lr=1e-3, L=2, K=19999, K_boundary=50,
                              loss_method="diffusion", verbose=False, device=dev, backend="native", noise="philox",
                              mlp_dtype=dt)
        m.V = psp.DenseNet(d_in=11, d_out=1, lr=1e-3, arch=[48, 48], seed=42).to(dev)
        m.train()
        res[dt] = (m.loss_log, m.K_log, m._gen_plan.grad.double().cpu())
    g32, g16 = res["fp32"][2], res["bf16"][2]
    cos = float(torch.dot(g32, g16) / (g32.norm() * g16.norm()))
    assert cos >= 0.9995, cos
    assert float((g32 - g16).abs().max()) <= 2e-2 * float(g32.abs().max())
    assert res["fp32"][1] == res["bf16"][1]
    for a, b in zip(res["bf16"][0], res["fp32"][0]):
        assert abs(a - b) <= 2e-2 * abs(b)
