"""The arithmetic claim behind k_gemm_bf16x6 (csrc/gcn.hip), checked in numpy on the CPU: an fp32 value splits EXACTLY into three bf16
terms (hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid, round to nearest even), and the three term products the kernel leaves out
(mid.lo, lo.mid, lo.lo) are together at most 2^-23 of the product (2 x 2^-8 x 2^-16: the size of one fp32 rounding step), in the mean 2^-28
and for the median operand pair 2^-29 -- and with round-to-nearest terms they carry no sign bias."""
import numpy as np


def bf16_rne(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 does for finite values)."""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x: np.ndarray):
    x = x.astype(np.float32)
    hi = bf16_rne(x)
    r1 = (x - hi).astype(np.float32)          # exact in fp32 (asserted below in float64)
    mid = bf16_rne(r1)
    lo = (r1 - mid).astype(np.float32)
    return hi, mid, lo


def samples(n=400_000, seed=0):
    rng = np.random.default_rng(seed)
    a = (rng.standard_normal(n) * np.exp2(rng.integers(-40, 40, n))).astype(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0**-23, 1.0 - 2.0**-24, 3.0, 0.1, 255.5, 2.0**-100, 65504.0, 1.9999999, 16777215.0,
                     np.float32(np.pi), np.float32(1e-30)], dtype=np.float32)     # (|x| >= 2^-100: a remainder must stay a normal number)
    return np.concatenate([a, edge])


def test_three_bf16_terms_restore_the_value_bit_for_bit():
    x = samples()
    hi, mid, lo = split3(x)
    assert np.array_equal(lo, bf16_rne(lo))                                             # the last remainder fits bf16: nothing is dropped
    total = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
    assert np.array_equal(total, x.astype(np.float64))                                  # hi + mid + lo == x exactly
    nz = x != 0
    assert np.all(np.abs(mid[nz].astype(np.float64)) <= 2.0**-8 * np.abs(x[nz].astype(np.float64)))
    assert np.all(np.abs(lo[nz].astype(np.float64)) <= 2.0**-16 * np.abs(x[nz].astype(np.float64)))
    # (round to nearest halves both: typical bounds 2^-9 and 2^-18; the asserted ones hold for every value incl. ties)


def test_the_three_products_left_out_are_of_the_size_of_one_fp32_rounding_or_below():
    a, b = samples(seed=1), samples(seed=2)
    ah, am, al = (t.astype(np.float64) for t in split3(a))
    bh, bm, bl = (t.astype(np.float64) for t in split3(b))
    exact = a.astype(np.float64) * b.astype(np.float64)
    kept = ah * bh + (ah * bm + am * bh) + (am * bm + ah * bl + al * bh)                # the six term products of k_gemm_bf16x6 (float64: exact sums here)
    nz = exact != 0
    rel = np.abs(exact[nz] - kept[nz]) / np.abs(exact[nz])
    assert rel.max() < 2.0**-23                                                         # hard bound: 2^-8 2^-16 + 2^-16 2^-8 + 2^-32 (measured max: 2^-24.2)
    assert np.quantile(rel, 0.999) < 2.0**-24 and rel.mean() < 2.0**-27.5 and np.median(rel) < 2.0**-28.5   # what round-to-nearest terms give in practice
    signed = (exact[nz] - kept[nz]) / np.abs(exact[nz])
    assert abs(signed.mean()) < 0.05 * rel.mean()                                       # no sign bias: the omissions average out over a dot product
    # every kept term product is exact in fp32: two 8-bit significands multiply to at most 16 bits
    for p, q in ((ah, bh), (ah, bm), (am, bh), (am, bm), (ah, bl), (al, bh)):
        prod = p * q
        fin = (np.abs(prod) < 3e38) & (np.abs(prod) > 1e-37)
        assert np.array_equal(prod[fin].astype(np.float32).astype(np.float64), prod[fin])
