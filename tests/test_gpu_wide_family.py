"""The two kernel families (narrow: hjb_kernels.h, wide: hjbw_kernels.h) on the SAME configuration.

The library picks the family per (d, H) when it is loaded (PSP_FORCE_WIDE=1 prefers the wide instance where both
exist), so each family runs in its own child process; the parent compares losses, per-trajectory D and the flat
gradient.  Tolerance: both are fp32 with different summation orders -> 2e-5 relative on D / loss, 2e-4 of max|grad|.
"""
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
prob = psp.LLGC(d=100, off_diag=0.01, T=0.3, seed=42, device=dev)
m = psp.Solver('fam', prob, lr=1e-3, L=3, K=%(K)d, delta_t=0.01, loss_method='log-variance', time_approx='inner',
               adaptive_forward_process=True, detach_forward=%(detach)s, u_l2_error_flag=False, verbose=False, seed=42,
               device=dev, backend='native', noise='philox', widths=(64, 64), mlp_dtype=%(mode)r)
m.train()
plan = m._native_plan
print(json.dumps({'family': psp.native.family(100, 64), 'loss': m.loss_log, 'D': plan.D.cpu().tolist()[:64],
                  'gmax': float(plan.grad.abs().max()), 'g': plan.grad.cpu().tolist()}))
"""


def run_child(force_wide, K, detach=True, mode='auto'):
    env = dict(os.environ)
    env['PSP_FORCE_WIDE'] = '1' if force_wide else '0'
    out = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, K=K, detach=detach, mode=mode)], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("K", [1000, 4096])      # ragged last tile / several workgroups
def test_wide_family_matches_narrow_family(K):
    a, b = run_child(False, K), run_child(True, K)
    assert a['family'] == 1 and b['family'] == 2
    for x, y in zip(a['loss'], b['loss']):
        assert abs(x - y) <= 2e-5 * max(1.0, abs(x)), (a['loss'], b['loss'])
    dmax = max(1.0, max(abs(v) for v in a['D']))
    assert max(abs(x - y) for x, y in zip(a['D'], b['D'])) <= 2e-5 * dmax
    assert max(abs(x - y) for x, y in zip(a['g'], b['g'])) <= 2e-4 * a['gmax']


@pytest.mark.parametrize("mode", ["fp32", "f16x3"])
def test_wide_adjoint_sweep_matches_narrow_family(mode):
    """Gradients through the state path (detach_forward=False: forward with the attached-mode image, adjoint sweep, backward) on
    both families, with the fp32-MFMA kernels and with the split-product forward + sweep (hjbw_adj_kernel<.., X3>)."""
    a, b = run_child(False, 2048, detach=False, mode=mode), run_child(True, 2048, detach=False, mode=mode)
    assert a['family'] == 1 and b['family'] == 2
    for x, y in zip(a['loss'], b['loss']):
        assert abs(x - y) <= 2e-5 * max(1.0, abs(x)), (a['loss'], b['loss'])
    assert max(abs(x - y) for x, y in zip(a['g'], b['g'])) <= 2e-4 * a['gmax']
