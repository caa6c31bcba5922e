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
