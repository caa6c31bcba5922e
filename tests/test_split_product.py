"""The arithmetic of the split-product kernels (csrc/hjb_kernels.h gemm_Tx), modelled in numpy: every operand x = hi + lo / 2048
with hi = f16(x), lo = f16((x - hi) * 2048), and a.b = hi.hi + (hi.lo + lo.hi) / 2048 with fp32 accumulation.  The claim the
kernels rest on: this product is as accurate as an fp32 product chain (error against fp64 within 1.3x of fp32's), over the
operand magnitudes the rollout sees, while a plain f16 product is three orders of magnitude worse.  (The GPU parity of the kernels
themselves is tests/test_gpu_split_product.py.)"""
import numpy as np
import pytest


def split(a):
    hi = a.astype(np.float16)
    lo = ((a - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return hi, lo


def product_errors(scale_w, scale_x, k=100, seed=0):
    rng = np.random.default_rng(seed)
    W = (rng.standard_normal((k, k)) * scale_w).astype(np.float32)
    X = (rng.standard_normal((k, 4096)) * scale_x).astype(np.float32)
    ref = W.astype(np.float64) @ X.astype(np.float64)
    f32 = W @ X
    wh, wl = split(W)
    xh, xl = split(X)
    up = lambda a: a.astype(np.float32)   # noqa: E731
    main = up(wh) @ up(xh)
    corr = up(wh) @ up(xl) + up(wl) @ up(xh)
    s3 = main + corr * np.float32(1.0 / 2048)
    n = np.abs(ref).mean()
    return np.abs(f32 - ref).mean() / n, np.abs(s3 - ref).mean() / n, np.abs(main - ref).mean() / n


@pytest.mark.parametrize("scale_w,scale_x", [(0.1, 1.0), (1e-4, 1.0), (1.0, 1e-3), (0.01, 0.1), (10.0, 3.0)])
def test_split_product_is_fp32_grade(scale_w, scale_x):
    e32, e3, e1 = product_errors(scale_w, scale_x)
    assert e3 <= 1.3 * e32, (e32, e3)
    assert e1 >= 500 * e32                         # a plain f16 product is not


def test_split_represents_an_operand_to_22_bits_across_the_f16_range():
    rng = np.random.default_rng(1)
    for mag in (1e-3, 1.0, 1e3, 6e4):
        x = (rng.uniform(0.5, 1.0, 10000) * mag * rng.choice([-1.0, 1.0], 10000)).astype(np.float32)
        hi, lo = split(x)
        back = hi.astype(np.float64) + lo.astype(np.float64) / 2048.0
        assert np.max(np.abs(back - x) / np.abs(x)) <= 2.0 ** -21


def test_tiny_operands_need_the_power_of_two_prescale():
    """Below the f16 normal range (6.1e-5) the hi part loses its bits: the reason hjb_bwd3_kernel and the split adjoint sweep scale
    the trajectory-weighted panels by a power of two first (exact) and scale their results back."""
    rng = np.random.default_rng(2)
    x = (rng.uniform(0.5, 1.0, 10000) * 3e-6).astype(np.float32)          # G ~ w sqrt(dt) xi with w ~ 1 / K
    hi, lo = split(x)
    raw = np.max(np.abs(hi.astype(np.float64) + lo.astype(np.float64) / 2048.0 - x) / x)
    gs = np.float32(2.0 ** 18)
    hi, lo = split(x * gs)
    scaled = np.max(np.abs((hi.astype(np.float64) + lo.astype(np.float64) / 2048.0) / float(gs) - x) / x)
    assert raw > 1e-6 and scaled <= 2.0 ** -21
