"""CPU: the arithmetic of the split-precision GEMM (whisper-burn_amd/csrc/gemm_f16x3.hip), restated in numpy and pinned.

x = hi + 2^-11 lo with hi = fp16(x), lo = fp16((x - hi) 2^11) keeps 22 of f32's 24 mantissa bits; a product of two fp16
values is exact in f32; the kernel sums hi.hi in one f32 accumulator and hi.lo + lo.hi in a second one that is scaled by
2^-11 at the end.  The properties the design rests on: the split reconstructs x to 2^-21 relative, its low part can never
overflow, the three-product sum is as close to the exact product as an f32 GEMM, and a bf16 split is a decade worse (which
is why the pieces are fp16).  tests/study_split_precision.py measures the same thing through the whole model."""
import numpy as np


def split_f16(x):
    hi = x.astype(np.float16).astype(np.float32)
    lo = ((x - hi) * np.float32(2048.0)).astype(np.float16).astype(np.float32)
    return hi, lo


def split_bf16(x):
    def bf(v):
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
        return u.astype(np.uint32).view(np.float32)
    hi = bf(x)
    return hi, bf(x - hi)


def test_the_split_reconstructs_f32_to_22_bits_and_cannot_overflow():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.float32(s) for s in (1e-3, 1.0, 50.0, 3e4)])
    x = np.clip(x, -65000, 65000).astype(np.float32)
    hi, lo = split_f16(x)
    assert np.isfinite(hi).all() and np.isfinite(lo).all()
    assert np.abs(lo).max() <= np.abs(x).max()                                             # |x - hi| 2^11 <= |x|: the low piece never overflows
    rec = hi.astype(np.float64) + lo.astype(np.float64) / 2048.0
    normal = np.abs(x) > 1e-3                                                              # (fp16 subnormal pieces: absolute, not relative)
    rel = np.abs(rec - x.astype(np.float64))[normal] / np.abs(x.astype(np.float64))[normal]
    assert rel.max() <= 2.0 ** -21, rel.max()


def test_three_fp16_products_are_as_close_to_the_exact_product_as_f32():
    rng = np.random.default_rng(11)
    M, K, N = 48, 1280, 96
    a = (rng.standard_normal((M, K)) * 1.5).astype(np.float32)              # LayerNorm-sized activations
    w = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    exact = a.astype(np.float64) @ w.astype(np.float64)
    f32 = a @ w
    ah, al = split_f16(a)
    wh, wl = split_f16(w)
    f16x3 = ah @ wh + (ah @ wl + al @ wh) * np.float32(1.0 / 2048.0)
    bh, bl = split_bf16(a)
    vh, vl = split_bf16(w)
    bf16x3 = bh @ vh + (bh @ vl + bl @ vh)
    e32, e16, eb = (np.abs(v.astype(np.float64) - exact).max() for v in (f32, f16x3, bf16x3))
    assert e16 <= 4.0 * e32 + 1e-7, (e16, e32)             # measured through the model: 0.9-1.4x (profiles/r03_split_precision_study.txt)
    assert eb >= 6.0 * e16, (eb, e16)                      # 16 mantissa bits instead of 22
    # the products themselves are exact in f32: 11-bit x 11-bit significands
    p = (ah[:, :64].astype(np.float64)[:, :, None] * wh[:64, :8].astype(np.float64)[None, :, :])
    assert (p.astype(np.float32).astype(np.float64) == p).all()
