"""Split-product mode (mlp_dtype='f16x3', PSP_MLP_F16X3): every matrix product of the forward rollout as three f16 MFMAs per fp32
product (hi/lo operand split, csrc/hjb_kernels.h gemm_Tx).  It is a mode of the fp32 path, so it is held to the SAME bounds as
the fp32 kernels: the oracle comparison of D and the gradient and the reference's golden loss logs of tests/test_gpu_parity.py,
plus a direct comparison of the two kernels' D on one Philox stream."""
import pytest
import torch

from conftest import load_golden
from util_cases import make_pkg_solver, psp
import test_gpu_parity as base

pytestmark = pytest.mark.gpu
nat = psp.native

# the narrow-family cases of the fp32 suite (the wide family keeps fp32 MFMA)
CASES = [c for c in base.NATIVE_CASES if not any(t in c for t in ("d200", "d500", "d300", "d105"))]


@pytest.mark.parametrize("name", CASES)
def test_split_first_iteration_D_and_gradient_match_oracle(name):
    base.check_first_iteration(name, mlp_dtype="f16x3")
    print("observed max |dD| / scale, max |dg| / max|g|:", base.check_first_iteration.observed)


@pytest.mark.parametrize("name", CASES)
def test_split_loss_log_matches_reference_golden(name):
    base.check_loss_log(name, mlp_dtype="f16x3")


@pytest.mark.parametrize("name,K", [("llgc_d100_h64_logvar", 4096), ("lqgc_d33_h50_logvar", 2048), ("dw_d70_h64_logvar", 2048)])
def test_split_D_tracks_fp32_kernel_on_philox_stream(name, K):
    """Same Philox stream, same weights, 100 steps: the per-trajectory D of the split-product kernel against the fp32 MFMA kernel.
    Both carry fp32 rounding noise of the same size against the exact result, so the bound is the parity bound on D."""
    case = load_golden(name)["case"]
    out = {}
    for mode in ("fp32", "f16x3"):
        m = make_pkg_solver(case, base.dev(), backend="native", noise="philox", L=1, K=K, mlp_dtype=mode)
        m.train()
        out[mode] = (m._native_plan.D.clone(), m.loss_log[0])
    scale = max(1.0, float(out["fp32"][0].abs().max()))
    err = float((out["fp32"][0] - out["f16x3"][0]).abs().max()) / scale
    print("max |D_f16x3 - D_fp32| / scale = %.3g" % err)
    assert err <= 2e-5
    assert abs(out["fp32"][1] - out["f16x3"][1]) <= 2e-5 * abs(out["fp32"][1])
