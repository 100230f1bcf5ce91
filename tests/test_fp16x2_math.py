"""The arithmetic claim behind RFE_OPT_LG_FP16X2 (rover-slam_amd/csrc/h2_split.h, gemm_h2.hip, lg_attention_h2.hip), checked in numpy -- no GPU:
every fp32 operand x = hi + lo with hi = fp16(x), lo = fp16(x - hi) carries 22 of x's 24 significand bits, and the three products
hi*hi + hi*lo + lo*hi summed in (at least) fp32 are as accurate as an fp32 fused-multiply-add chain.  The GPU tests measure the same thing on the
kernels (tests/test_gpu_attention.py, tools/kbench/gemm_bf16x3.hip -> profiles/r03_fp16x2_kbench.md); this one pins the bound itself."""
import numpy as np


def _split(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)      # the residual is exact in fp32 (Sterbenz-like: |x - hi| <= ulp16(x) / 2)
    return hi, lo


def test_split_carries_22_bits():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp2(rng.integers(-10, 11, 200000))).astype(np.float32)   # |x| from ~1e-4 to ~1e4
    hi, lo = _split(x)
    err = np.abs(x.astype(np.float64) - (hi.astype(np.float64) + lo.astype(np.float64)))
    # 2^-22 relative (two round-to-nearest fp16 terms: 11 + 11 bits) plus fp16's absolute floor for residuals in the subnormal range
    assert (err <= np.abs(x).astype(np.float64) * 2.0 ** -22 + 2.0 ** -25).all(), float((err / np.abs(x)).max())
    assert np.isfinite(hi.astype(np.float32)).all() and np.isfinite(lo.astype(np.float32)).all()


def test_three_products_are_fp32_class():
    """Dot products of LightGlue's ffn.0 shape (K = 512; activations O(1) against weights O(0.05)): error against float64 relative to
    sum |a||b|, split products (exact fp16 x fp16 products, fp32 accumulation in blocks of 16 like the MFMA, then fp32 adds) beside a plain
    fp32 fmaf-style chain."""
    rng = np.random.default_rng(1)
    rows, K = 4096, 512
    a = rng.standard_normal((rows, K)).astype(np.float32)
    b = (0.05 * rng.standard_normal((rows, K))).astype(np.float32)
    exact = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    scale = (np.abs(a).astype(np.float64) * np.abs(b).astype(np.float64)).sum(1)
    ah, al = _split(a)
    bh, bl = _split(b)
    f64 = lambda t: t.astype(np.float64)
    # products of two fp16 numbers are exact in fp32; a 32x32x16 MFMA sums 16 of them and the accumulator in fp32
    prod = (f64(al) * f64(bh) + f64(ah) * f64(bl) + f64(ah) * f64(bh)).astype(np.float32)     # per-k sum of the three terms, rounded to fp32
    acc = np.zeros(rows, np.float32)
    for k0 in range(0, K, 16):
        acc = (acc + prod[:, k0:k0 + 16].sum(1, dtype=np.float32)).astype(np.float32)
    err_split = np.abs(acc.astype(np.float64) - exact) / scale
    chain = np.zeros(rows, np.float32)
    for k in range(K):
        chain = (chain.astype(np.float64) + a[:, k].astype(np.float64) * b[:, k].astype(np.float64)).astype(np.float32)   # fmaf: one rounding per step
    err_chain = np.abs(chain.astype(np.float64) - exact) / scale
    rms = lambda e: float(np.sqrt((e ** 2).mean()))
    print(f"split: rms {rms(err_split):.2e} max {err_split.max():.2e}   fp32 chain: rms {rms(err_chain):.2e} max {err_chain.max():.2e}")
    assert rms(err_split) < 1.5 * rms(err_chain) + 1e-8 and err_split.max() < 1e-6
    # two bf16 terms and three products -- the cheaper candidate that was rejected -- are an order of magnitude worse
    def bf16(t):
        u = t.astype(np.float32).view(np.uint32)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
        return r.view(np.float32)
    a0, b0 = bf16(a), bf16(b)
    a1, b1 = bf16(a - a0), bf16(b - b0)
    acc_bf = (f64(a0) * f64(b0) + f64(a0) * f64(b1) + f64(a1) * f64(b0)).sum(1)
    err_bf = np.abs(acc_bf - exact) / scale
    assert rms(err_bf) > 5 * rms(err_split)
