"""The fused attention kernels alone (rfe_k_attention) against a float64 evaluation: softmax(Q K^T / 8) V per (sequence, head), with
LightGlue's rotary on q / k (self blocks) or a kv_map (cross blocks), ragged lengths, at the THROUGHPUT shape (32 sequences of 1024 =
32 768 query rows).  Both arithmetic routes of the throughput path are measured against the same float64 result:
  * the default fp32-MFMA kernels (lg_attention_kernel / lg_attention_dma_kernel, lg_kernels.hip),
  * RFE_OPT_LG_FP16X2 = 1: lg_attention_h2.hip (fp16 hi + lo operand pairs, three products, fp32 accumulation),
and the split route may not be less accurate than fp32-class: the bound is the same for both.  The reference evaluates this inside
Session::Run(lightglue_sim.onnx), src/Matchers/lightglue_onnx.cpp:210-214."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATT_TOL = 2e-5   # max |context - float64| for O(1) values; the fp32 kernels sit at ~1e-6


@pytest.fixture(scope="module")
def ctx():
    from rover_slam_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _attention_f64(q, k, v, nq, nk, rope_q=None, rope_k=None):
    """q [Lq,256], k / v [Lk,256] float32 -> context [Lq,256] float64; rows >= nq are zero."""
    def rot(x, cs):
        if cs is None:
            return x.astype(np.float64)
        x = x.astype(np.float64).reshape(x.shape[0], 4, 32, 2)
        c, s = cs[:, None, :, 0].astype(np.float64), cs[:, None, :, 1].astype(np.float64)
        a, b = x[..., 0], x[..., 1]
        return np.stack([a * c - b * s, b * c + a * s], -1).reshape(x.shape[0], 256)
    qq, kk, vv = rot(q, rope_q)[:nq], rot(k, rope_k)[:nk], v.astype(np.float64)[:nk]
    out = np.zeros((q.shape[0], 256))
    for h in range(4):
        s = qq[:, 64 * h:64 * h + 64] @ kk[:, 64 * h:64 * h + 64].T * 0.125
        s -= s.max(1, keepdims=True)
        p = np.exp(s)
        p /= p.sum(1, keepdims=True)
        out[:nq, 64 * h:64 * h + 64] = p @ vv[:, 64 * h:64 * h + 64]
    return out


def _run(ctx, capi, x, offs, ld, nseq, L, lens, kvmap, rope, fp16x2):
    bufs = []
    def up(a):
        b = ctx.alloc(a.nbytes); b.upload(a); bufs.append(b); return b
    dq = up(x)
    dout = ctx.alloc(nseq * L * 1024); bufs.append(dout)
    dlen = up(lens.astype(np.int32))
    dmap = up(kvmap.astype(np.int32)) if kvmap is not None else None
    drope = up(rope) if rope is not None else None
    ctx.set_option(capi.OPT_LG_FP16X2, 1 if fp16x2 else 0)
    try:
        ctx._chk(capi.lib.rfe_k_attention(ctx.h, dq.ptr + offs[0] * 4, dq.ptr + offs[1] * 4, dq.ptr + offs[2] * 4, ld, dout.ptr, nseq, L, L, dlen.ptr, dlen.ptr,
                                          dmap.ptr if dmap else None, drope.ptr if drope else None))
    finally:
        ctx.set_option(capi.OPT_LG_FP16X2, 0)
    out = dout.download((nseq, L, 256), np.float32)
    for b in bufs:
        b.free()
    return out


@pytest.mark.parametrize("kind", ["self_rope", "cross", "self_rope_L1000", "cross_L1000"])
def test_attention_throughput_shape_vs_float64(ctx, kind):
    from rover_slam_amd import capi
    rng = np.random.default_rng({"cross": 5, "self_rope": 6, "cross_L1000": 7, "self_rope_L1000": 8}[kind])
    nseq, L = (34, 1000) if kind.endswith("L1000") else (32, 1024)   # L = 1000: a padded length that is a multiple of 4 only (not of the 64-key tile / 256-query block)
    kind = kind.replace("_L1000", "")
    lens = np.full(nseq, L, np.int32)
    lens[[1, 6, 7, 20]] = [650, min(1001, L - 3), 257, 32]          # ragged: partial last tiles, a sequence shorter than one query block
    if kind == "self_rope":
        ld, offs, kvmap = 768, (0, 256, 512), None       # [q | k | v] rows as the self block's projection writes them
        rope = np.empty((nseq * L, 32, 2), np.float32)
        th = rng.uniform(-3.0, 3.0, (nseq * L, 32))
        rope[..., 0], rope[..., 1] = np.cos(th), np.sin(th)
    else:
        ld, offs, rope = 512, (0, 0, 256), None          # [qk | v]: the cross block's shared query / key projection
        kvmap = np.arange(nseq) ^ 1                      # sequence 2 p attends to 2 p + 1 and back
    x = rng.standard_normal((nseq * L, ld)).astype(np.float32)
    x[:, :ld - 256] *= 1.5                               # q / k: logits with a standard deviation of ~2.3, peaky rows
    x[::7, 3] += 6.0                                     # a few dominant keys / queries: exercises the moving softmax reference
    xs = x.reshape(nseq, L, ld)
    check = [0, 1, 6, 7, 20, 21, 31]
    ref = {}
    for s in check:
        t = kvmap[s] if kvmap is not None else s
        r = rope.reshape(nseq, L, 32, 2) if rope is not None else None
        ref[s] = _attention_f64(xs[s][:, offs[0]:offs[0] + 256], xs[t][:, offs[1]:offs[1] + 256], xs[t][:, offs[2]:offs[2] + 256],
                                int(lens[s]), int(lens[t]), r[s] if r is not None else None, r[t] if r is not None else None)
    errs = {}
    for fp16x2 in (0, 1):
        out = _run(ctx, capi, x, offs, ld, nseq, L, lens, kvmap, rope, fp16x2)
        assert np.isfinite(out).all()
        worst = 0.0
        for s in check:
            worst = max(worst, float(np.abs(out[s] - ref[s]).max()))
            assert not out[s][lens[s]:].any(), "context rows past the sequence length must be zero"
        errs[fp16x2] = worst
    print(f"attention {kind}: max |context - float64|  fp32 kernels {errs[0]:.2e}   fp16x2 split kernel {errs[1]:.2e}")
    assert errs[0] < ATT_TOL and errs[1] < ATT_TOL, errs


@pytest.mark.parametrize("fp16x2", [0, 1])
@pytest.mark.parametrize("kind,nseq,L", [("self", 2, 1024), ("cross", 2, 1024), ("cross", 6, 152), ("self", 4, 1000), ("cross", 8, 1024), ("cross", 2, 36)])
def test_attention_latency_shape_vs_float64(ctx, kind, nseq, L, fp16x2):
    """The shapes the reference itself runs (one / few pairs per call, src/Matchers/lightglue_onnx.cpp:168-172): at most 8192 query rows and
    no rotary table (the one-pair projection has rotated q and k) take lg_attention_lat_kernel -- the key split inside the workgroup,
    self-pipelined waves, partial sums merged through LDS.  Ragged lengths put partial tiles first, last and alone in a wave's key
    quarter, leave waves without keys and sequences shorter than one tile.  fp16x2 = 1: the same kernel's split form (RFE_OPT_LG_FP16X2:
    K, Q, V and P as fp16 hi + lo, three v_mfma_f32_32x32x16_f16 per block), same bar."""
    from rover_slam_amd import capi
    rng = np.random.default_rng(nseq * 1000 + L)
    lens = np.full(nseq, L, np.int32)
    ragged = [L - 3, 131, 70, 40, 33, 3, 97, 1]
    for i in range(1, nseq):
        lens[i] = min(L, ragged[(i - 1) % len(ragged)])
    if kind == "self":
        ld, offs, kvmap = 768, (0, 256, 512), None
    else:
        ld, offs, kvmap = 512, (0, 0, 256), np.arange(nseq) ^ 1
    x = rng.standard_normal((nseq * L, ld)).astype(np.float32)
    x[:, :ld - 256] *= 1.5
    x[::7, 3] += 6.0                                     # dominant keys / queries: the softmax reference moves late in a key range
    x[L // 2::11, 5] += 9.0
    xs = x.reshape(nseq, L, ld)
    out = _run(ctx, capi, x, offs, ld, nseq, L, lens, kvmap, None, fp16x2)
    assert np.isfinite(out).all()
    worst = 0.0
    for s in range(nseq):
        t = kvmap[s] if kvmap is not None else s
        ref = _attention_f64(xs[s][:, offs[0]:offs[0] + 256], xs[t][:, offs[1]:offs[1] + 256], xs[t][:, offs[2]:offs[2] + 256], int(lens[s]), int(lens[t]))
        worst = max(worst, float(np.abs(out[s] - ref).max()))
        assert not out[s][lens[s]:].any(), "context rows past the sequence length must be zero"
    print(f"attention latency shape {kind} nseq={nseq} L={L} fp16x2={fp16x2}: max |context - float64| {worst:.2e}")
    assert worst < ATT_TOL


def test_attention_split_kernel_saturates_outside_fp16_range(ctx):
    """RFE_OPT_LG_FP16X2's domain is |operand| < 65504 (fp16).  Outside it the split kernels run with MODE.FP16_OVFL = 1 (h2_split.h): hi
    saturates at 65504, lo takes up the rest -- values up to 131008 are still carried to fp16's 11 bits -- instead of inf - inf = NaN for
    every query that attends to the offending key.  V entries of +-9e4 here: finite everywhere and within 1e-3 relative of float64."""
    from rover_slam_amd import capi
    rng = np.random.default_rng(11)
    nseq, L, ld, offs = 32, 1024, 768, (0, 256, 512)
    x = rng.standard_normal((nseq * L, ld)).astype(np.float32)
    big = rng.integers(0, nseq * L, 64)
    x[big, 512 + rng.integers(0, 256, 64)] = rng.choice([-9.0e4, 9.0e4], 64).astype(np.float32)
    lens = np.full(nseq, L, np.int32)
    out = _run(ctx, capi, x, offs, ld, nseq, L, lens, None, None, 1)
    assert np.isfinite(out).all(), "the split attention produced non-finite context rows"
    xs = x.reshape(nseq, L, ld)
    for s in (0, 13, 31):
        ref = _attention_f64(xs[s][:, :256], xs[s][:, 256:512], xs[s][:, 512:], L, L)
        scale = np.abs(ref).max()
        assert np.abs(out[s] - ref).max() < 1e-3 * scale, (s, float(np.abs(out[s] - ref).max()), float(scale))
