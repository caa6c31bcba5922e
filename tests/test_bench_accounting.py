"""bench.py's flop / MFMA accounting against the counters measured on the MI355X (profiles/traffic.json, written by
tools/pmc_traffic.sh + tools/make_traffic_json.py from rocprofv3 --pmc SQ_INSTS_MFMA passes): the `frac_issued` figure of the
roofline object rests on `issued_mfma_per_tile_step`, which must reproduce the measured instruction counts exactly."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _traffic():
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
        return json.load(fh)


# workload -> (padded d, H, dense, kernel family, forward kernel name, backward kernel name)
CASES = {
    "hjb_llgc_d100_K65536_N100_h64_fp32mfma": (100, 64, True, 1, "hjb_fwd_kernel", "hjb_bwd2_kernel"),
    "hjb_llgc_d100_K1024_N50_h64": (100, 64, True, 1, "hjbq_fwd_kernel", "hjb_bwd2_kernel"),
    "hjb_llgc_d200_K32768_N100_h64_fp32mfma": (200, 64, True, 2, "hjbw_fwd_kernel", "hjbw_bwd2_kernel"),
    "hjb_llgc_d500_K16384_N200_h64_fp32mfma": (500, 64, True, 2, "hjbw_fwd_kernel", "hjbw_bwd_kernel"),
}
# the default workloads run the split-product kernels: (padded d, H, family, forward kernel, backward kernel or None)
X3_CASES = {
    "hjb_llgc_d100_K65536_N100_h64": (100, 64, 1, "hjb_fwd_kernel", "hjb_bwd3_kernel"),
    "hjb_llgc_d200_K32768_N100_h64": (200, 64, 2, "hjbw_fwd_kernel", None),
    "hjb_llgc_d500_K16384_N200_h64": (500, 64, 2, "hjbw_fwd_kernel", None),
}


@pytest.mark.parametrize("workload", sorted(X3_CASES))
def test_issued_mfma_formula_of_the_split_kernels_matches_the_pmc_counts(workload):
    d, H, family, fwd_name, bwd_name = X3_CASES[workload]
    meas = _traffic()[workload]
    w = bench.WORKLOADS[workload]
    N = int(round(w["T"] / w["dt"]))
    tiles_steps = (w["K"] // 16) * N
    fwd, bwd = bench.issued_mfma_x3(d, H, True, family)
    assert sum(fwd.values()) * tiles_steps == pytest.approx(meas[fwd_name]["mfma_instructions"], rel=2e-3), (fwd, meas[fwd_name])
    if bwd_name:
        assert sum(bwd.values()) * tiles_steps == pytest.approx(meas[bwd_name]["mfma_instructions"], rel=2e-3), (bwd, meas[bwd_name])


@pytest.mark.parametrize("workload", sorted(CASES))
def test_issued_mfma_formula_matches_the_pmc_counts(workload):
    d, H, dense, family, fwd_name, bwd_name = CASES[workload]
    meas = _traffic()[workload]
    w = bench.WORKLOADS[workload]
    N = int(round(w["T"] / w["dt"]))
    tiles_steps = (w["K"] // 16) * N
    fwd, _, bwd = bench.issued_mfma_per_tile_step(d, H, dense, family)
    if fwd_name == "hjbq_fwd_kernel":           # 4x4x1 instructions: four per 16x16x4 equivalent
        fwd = 4 * bench.issued_mfma_quad_kernel(d, H, dense)
    assert fwd * tiles_steps == pytest.approx(meas[fwd_name]["mfma_instructions"], rel=2e-3), (fwd, meas[fwd_name])
    assert bwd * tiles_steps == pytest.approx(meas[bwd_name]["mfma_instructions"], rel=2e-3), (bwd, meas[bwd_name])


def test_minimal_flops_are_below_issued_flops_and_survey_m3_above_minimal():
    for d in (100, 200, 500):
        fl2 = bench.alg_flops_per_traj_step(d, 64, True, m=2)
        fl3 = bench.alg_flops_per_traj_step(d, 64, True, m=3)
        fwd, _, bwd = bench.issued_mfma_per_tile_step(d, 64, True, 2 if d > 112 else 1)
        assert fl3["fwd_kernel"] - fl2["fwd_kernel"] == 2 * d * d          # one dense d x d product
        assert fl2["fwd_kernel"] <= fwd * bench.MFMA_F32_16x16x4_FLOP / 16.0 + 12 * d   # issued >= minimal (padding to 16-blocks)
        assert fl2["bwd_kernel"] <= bwd * bench.MFMA_F32_16x16x4_FLOP / 16.0


def test_path_store_traffic_equals_the_algorithmic_store():
    """WRITE_SIZE of the forward kernels is the path store to within 0.5 % (nothing else is written), and the calibrated read
    traffic of the backward kernels lies between one and two passes over it."""
    t = _traffic()
    store = 1408 * 65536 * 100                                              # d=100, H=64: X_n, xi, h1, h2 images
    for wl in ("hjb_llgc_d100_K65536_N100_h64", "hjb_llgc_d100_K65536_N100_h64_fp32mfma"):
        assert t[wl]["hjb_fwd_kernel"]["write_bytes"] == pytest.approx(store, rel=5e-3)
    b = t["hjb_llgc_d100_K65536_N100_h64_fp32mfma"]["hjb_bwd2_kernel"]
    assert store <= 2 * b["fetch_raw_bytes"] <= 2 * store
    bf = t["diffusion_dw_d100_K65536_N100_h64_bf16"]["gen_fwd_kernel"]
    assert bf["write_bytes"] == pytest.approx(960 * 65536 * 101 + 4 * 65536 * 101, rel=2e-2)   # bf16-pair images + ahat


def test_quad_kernel_mfma_count():
    """hjbq_fwd_kernel at the configs[1] shape: 104 v_mfma_f32_4x4x1 per wave and step (W1 16, dt A 32, B 32, W2 8, W3 16),
    8 waves, 4 quads per 16 trajectories, 512 flop each = 832 16x16x4 equivalents; without dense drift / sigma 40 per wave."""
    assert bench.issued_mfma_quad_kernel(100, 64, True) == 832
    assert bench.issued_mfma_quad_kernel(100, 64, False) == 8 * 40
    assert bench.issued_mfma_quad_kernel(16, 16, True) == 8 * (4 * 3 + 2 * 2)
