"""The arithmetic behind the opt-in pipe k_gemm_f16x3 (csrc/gcn.hip "F16x3", MDFRI_HW_PIPE=f16x3), checked in numpy on the CPU: an fp32 value
times a power of two splits into TWO fp16 terms (hi = f16(x s), lo = f16(x s - hi), round to nearest even) whose sum holds the value to
2^-22; the remainder x s - hi is exact in fp32 (what lets one v_fma_mix_f32 form it); the one term product the kernel leaves out plus the two
operand remainders are together at most 3 x 2^-22 of the product, an order below in the mean and without sign bias; and the scales: a value
whose magnitude times the scale is below 2^-2 keeps an ABSOLUTE accuracy of 2^-25 / s instead (its low term is a subnormal fp16), one whose
scaled magnitude exceeds 65 504 becomes inf -- the two limits include/mdfri.h states for the activations' constant scale 2^3."""
import numpy as np

SCALE_A = 8.0    # F16X3_SCALE_A (csrc/gcn.hip)


def split2(x: np.ndarray, s: float):
    y = (x.astype(np.float32) * np.float32(s)).astype(np.float32)      # a power of two: exact
    with np.errstate(over="ignore"):
        hi = y.astype(np.float16)                                       # round to nearest even (v_cvt_pk_f16_f32)
    rem64 = y.astype(np.float64) - hi.astype(np.float64)
    rem = rem64.astype(np.float32)
    lo = rem.astype(np.float16)
    return y, hi, rem64, rem, lo


def weight_scale(wmax: float) -> float:
    """The scale mdf_model_create derives from max |W| (csrc/gcn.hip: Wt_scale)."""
    return float(np.ldexp(1.0, 14 - int(np.ceil(np.log2(wmax))))) if wmax > 0 and np.isfinite(wmax) else 1.0


def samples(n=300_000, seed=0, lo_exp=-5, hi_exp=9):
    rng = np.random.default_rng(seed)
    return (rng.uniform(1.0, 2.0, n) * np.exp2(rng.integers(lo_exp, hi_exp + 1, n)) * rng.choice([-1.0, 1.0], n)).astype(np.float32)


def test_two_fp16_terms_hold_a_scaled_value_to_2_pow_minus_22_and_the_remainder_is_exact():
    x = samples()                                                       # 2^-5 <= |x| < 2^10: the range the constant scale keeps at full accuracy
    y, hi, rem64, rem, lo = split2(x, SCALE_A)
    assert np.all(np.isfinite(hi.astype(np.float32)))
    assert np.array_equal(rem.astype(np.float64), rem64)                # x s - hi is representable in fp32: the fma returns it unrounded
    assert np.all(np.abs(rem64) <= 2.0**-11 * np.abs(y.astype(np.float64)))
    err = np.abs(y.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64))
    assert np.all(err <= 2.0**-22 * np.abs(y.astype(np.float64)))      # 11 + 11 bits
    assert err.mean() < 2.0**-24.5 * np.abs(y.astype(np.float64)).mean()
    # (a low term below 2^-14 is a subnormal fp16 and holds fewer than 11 bits: with |x s| >= 2^-2 the 2^-22 bound above still holds for
    # every sample, because the remainder's own last bit is 2^-23 |x s| >= 2^-25, the subnormal step 2^-24, and half of it is the error)


def test_the_product_left_out_and_the_remainders_stay_below_3_x_2_pow_minus_22():
    a, b = samples(seed=1), samples(seed=2, lo_exp=-9, hi_exp=-2)      # activations; weights
    sb = weight_scale(float(np.abs(b).max()))
    ya, ah, _, _, al = split2(a, SCALE_A)
    yb, bh, _, _, bl = split2(b, sb)
    ah, al, bh, bl = (t.astype(np.float64) for t in (ah, al, bh, bl))
    exact = ya.astype(np.float64) * yb.astype(np.float64)
    kept = (al * bh + ah * bl) + ah * bh                                # the three term products of k_gemm_f16x3, smallest first
    rel = np.abs(exact - kept) / np.abs(exact)
    assert rel.max() < 3 * 2.0**-22
    assert rel.mean() < 2.0**-23.5 and np.median(rel) < 2.0**-24
    signed = (exact - kept) / np.abs(exact)
    assert abs(signed.mean()) < 0.05 * rel.mean()                       # no sign bias
    for p, q in ((al, bh), (ah, bl), (ah, bh)):                         # every kept term product is exact in fp32: 11 x 11 bits
        prod = p * q
        assert np.array_equal(prod.astype(np.float32).astype(np.float64), prod)
    # taking the scales out again is exact
    assert np.array_equal((np.float32(3.25) * np.float32(SCALE_A * sb)) * np.float32(1.0 / (SCALE_A * sb)), np.float32(3.25))


def test_weight_scale_brings_the_maximum_to_the_top_of_the_fp16_range_without_overflow():
    for wmax in (1e-6, 0.0442, 0.208, 0.25, 0.9999, 1.0, 3.7, 64.0, 1000.0):
        s = weight_scale(wmax)
        assert np.log2(s) == int(np.log2(s)) and 2.0**13 <= wmax * s <= 2.0**14 < 65504.0, (wmax, s)
    assert weight_scale(0.0) == 1.0 and weight_scale(float("inf")) == 1.0


def test_the_limits_of_the_constant_activation_scale():
    # below 2^-5: the low term is a subnormal fp16 (or zero), the value is held to 2^-25 of the SCALED unit = 2^-28 absolute
    tiny = samples(seed=3, lo_exp=-30, hi_exp=-6)
    y, hi, _, _, lo = split2(tiny, SCALE_A)
    err = np.abs(y.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64)) / SCALE_A
    assert err.max() <= 2.0**-28
    # at and beyond 65 520 / 8 = 8 190 the high term is inf (the kernel's products then give NaN scores: loud, not silently wrong)
    edge = np.array([8187.9, 8190.0, 9000.0, -1e6], dtype=np.float32)
    _, hi, _, _, _ = split2(edge, SCALE_A)
    assert np.isfinite(hi[0]) and np.all(np.isinf(hi[1:].astype(np.float32)))
