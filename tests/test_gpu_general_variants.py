"""GeneralSolver backward: the role-specialised kernel (gen_bwd2_kernel) against the two-workgroups-per-CU kernel
(gen_bwd_kernel, PSP_BWD_VARIANT=1) on a problem large enough that every workgroup runs MANY rounds (the golden cases
are one round per workgroup, which cannot see cross-round pipeline bugs).  The variant is fixed when the library is
loaded, so each runs in a child process.  Tolerance: fp32 with different summation orders -> 2e-4 of max|grad|."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, torch
sys.path.insert(0, %(root)r)
import path_space_pde_solver_amd as psp
dev = torch.device('cuda:0')
from path_space_pde_solver_amd import problems
prob = problems.DoubleWell_multidim_for_general_solver(d=10, d_1=5, d_2=5, T=0.3, eta=1.0, kappa=1.0, modus='HJB', device=dev)
m = psp.GeneralSolver(prob, 'var', seed=42, delta_t=0.01, N=20, lr=1e-3, L=2, K=%(K)d, K_boundary=50,
                      loss_method='%(loss)s', verbose=False, device=dev, backend='native', noise='philox')
m.V = psp.DenseNet(d_in=11, d_out=1, lr=1e-3, arch=[%(H)d, %(H)d], seed=42).to(dev)
m.train()
plan = m._gen_plan
print(json.dumps({'loss': m.loss_log, 'K_log': m.K_log, 'gmax': float(plan.grad.abs().max()), 'g': plan.grad.cpu().tolist()}))
"""


def run_child(variant, K, H, loss):
    env = dict(os.environ)
    env['PSP_BWD_VARIANT'] = variant
    out = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, K=K, H=H, loss=loss)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("K,H,loss", [(32768, 24, 'diffusion'), (20000, 32, 'BSDE')])
def test_role_specialised_backward_matches_two_workgroup_backward(K, H, loss):
    a, b = run_child('1', K, H, loss), run_child('0', K, H, loss)
    assert a['K_log'] == b['K_log']
    for x, y in zip(a['loss'], b['loss']):
        assert abs(x - y) <= 2e-5 * max(1.0, abs(x)), (a['loss'], b['loss'])
    assert max(abs(x - y) for x, y in zip(a['g'], b['g'])) <= 2e-4 * a['gmax']


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
    for name in ("dwgen_d10_diffusion", "allencahn_d20_default_diffusion", "dwgen_d40_h50_bsde", "heat_d6_diffusion"):
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
