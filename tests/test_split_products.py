"""The product form of the LDS-tiled dense kernels (glass_amd/csrc/split_mma.h): an fp32 operand cut into three bf16 pieces,
a product formed from six partial products on the bf16 matrix cores.  The arithmetic the claim rests on, restated in numpy;
the kernels themselves in both product forms against fp64: tests/test_gpu_kernels.py::test_both_product_forms_against_fp64."""
import numpy as np


def bf16_rne(x):
    """fp32 -> the nearest bf16 (ties to even), returned as fp32 (v_cvt_pk_bf16_f32's rounding for finite values)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    hi = bf16_rne(x)
    r1 = x - hi
    mid = bf16_rne(r1)
    r2 = r1 - mid
    lo = bf16_rne(r2)
    return hi, mid, lo, r2 - lo


def test_three_bf16_pieces_hold_an_fp32_value_exactly():
    rng = np.random.default_rng(0)
    x = np.concatenate([
        rng.standard_normal(200000).astype(np.float32),
        (rng.standard_normal(200000) * np.exp(8 * rng.standard_normal(200000))).astype(np.float32),
        # every exponent from 2^-100 up, random mantissas (below ~2^-109 the low pieces reach the denormal range and the cut
        # is no longer exact — magnitudes no activation, weight or gradient of this path has)
        rng.integers(27 << 23, 0x7F000000, 200000, dtype=np.uint32).view(np.float32),
        -rng.integers(27 << 23, 0x7F000000, 100000, dtype=np.uint32).view(np.float32),
        np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0**-23, 1.0 - 2.0**-24, 3.0e38, 2.0**-100, 16777215.0, 0.1], dtype=np.float32)])
    hi, mid, lo, rest = split3(x)
    assert np.all(rest == 0.0)                              # nothing left after three pieces
    assert np.all((hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)) == x.astype(np.float64))
    big = np.abs(x) > 2.0**-100                             # (relative sizes; far from the denormal range)
    assert np.all(np.abs(mid[big]) <= 2.0**-8 * np.abs(x[big])) and np.all(np.abs(lo[big]) <= 2.0**-16 * np.abs(x[big]))


def test_six_partial_products_error_below_one_fp32_rounding():
    """sum of the six kept partial products (exact, in fp64) against the exact product: the three dropped terms stay below
    2^-24 |x w|, one rounding of an fp32 product."""
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(300000) * np.exp(3 * rng.standard_normal(300000))).astype(np.float32)
    w = (rng.standard_normal(300000) * np.exp(3 * rng.standard_normal(300000))).astype(np.float32)
    xh, xm, xl, _ = (a.astype(np.float64) for a in split3(x))
    wh, wm, wl, _ = (a.astype(np.float64) for a in split3(w))
    kept = xm * wm + xl * wh + xh * wl + xm * wh + xh * wm + xh * wh
    exact = x.astype(np.float64) * w.astype(np.float64)
    rel = np.abs(kept - exact) / np.abs(exact)
    assert rel.max() < 2.0**-24, rel.max()
