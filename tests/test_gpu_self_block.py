"""The self block's projection + attention alone (rfe_k_lightglue_self_attention = the forward's own lg_self_qkv_attention) against float64:
  q | k | v = Wqkv x + b,  q and k rotated pair-wise by the rotary table,  context = softmax(q k^T / 8) v per head.
The row count selects the code under test, as inside a match call:
  * throughput shapes: the rotary runs in the qkv projection's EPILOGUE (gemm.hip, ROPE tile: pair partner by DPP, table tile staged into LDS by
    global_load_lds) and the attention is lg_attention_dma_kernel without a table (round 5);
  * one pair: gemm_lat.hip's rotary epilogue + lg_attention_lat.hip;
  * shapes neither takes: plain epilogue + lg_attention_kernel<.., ROPE> rotating on load.
The reference evaluates all of this inside Session::Run(lightglue_sim.onnx), src/Matchers/lightglue_onnx.cpp:210-214; the oracle's statement is
oracle/rfe_oracle.c (rfo_lightglue: rotary, self attention)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

QKV_TOL = 2e-5    # |q k v - float64| for O(1) projections (K = 256 fp32 dot products: ~2e-6 observed)
CTX_TOL = 2e-5    # as tests/test_gpu_attention.py


@pytest.fixture(scope="module")
def ctx_w():
    from rover_slam_amd import capi, weights as Wt
    c = capi.Context(0)
    w = Wt.make_lightglue(seed=11, calibrated=True)
    c.set_weights(capi.KIND_LIGHTGLUE, w)
    yield c, w
    c.close()


def _f64(x, W, b, cs, lens, L):
    qkv = x.astype(np.float64) @ W.astype(np.float64).T + b.astype(np.float64)
    rot = qkv.copy()
    c, s = cs[:, None, :, 0].astype(np.float64), cs[:, None, :, 1].astype(np.float64)
    for o in (0, 256):
        t = qkv[:, o:o + 256].reshape(-1, 4, 32, 2)
        a, bb = t[..., 0], t[..., 1]
        rot[:, o:o + 256] = np.stack([a * c - bb * s, bb * c + a * s], -1).reshape(-1, 256)
    out = np.zeros((x.shape[0], 256))
    for sq, n in enumerate(lens):
        r = rot[sq * L:sq * L + n]
        for h in range(4):
            sc = r[:, 64 * h:64 * h + 64] @ r[:, 256 + 64 * h:256 + 64 * h + 64].T * 0.125
            sc -= sc.max(1, keepdims=True)
            p = np.exp(sc)
            p /= p.sum(1, keepdims=True)
            out[sq * L:sq * L + n, 64 * h:64 * h + 64] = p @ r[:, 512 + 64 * h:512 + 64 * h + 64]
    return qkv, rot, out


@pytest.mark.parametrize("nseq,L,expect_rotated", [(12, 1024, 1),     # 12 288 rows: throughput tile with the rotary epilogue (>= 256 tiles of 128 x 256)
                                                   (11, 1000, 1),     # 11 000 rows, not a multiple of 128: the last row panel is partial (clamped table rows)
                                                   (2, 1024, 1),      # one pair: gemm_lat.hip's rotary epilogue
                                                   (6, 512, None)])   # few pairs: whichever path serves it -- the result must be right either way
def test_self_block_projection_and_attention_vs_float64(ctx_w, nseq, L, expect_rotated):
    from rover_slam_amd import capi, weights as Wt
    ctx, w = ctx_w
    man = {name: (off, shape) for name, off, shape in Wt.lg_manifest()[0]}
    layer = 3
    off, shp = man[f"layers.{layer}.self.Wqkv"]; W = w[off:off + 768 * 256].reshape(768, 256)
    off, _ = man[f"layers.{layer}.self.bqkv"]; b = w[off:off + 768]
    rng = np.random.default_rng(100 + nseq)
    rows = nseq * L
    x = rng.standard_normal((rows, 256)).astype(np.float32)
    th = rng.uniform(-3.0, 3.0, (rows, 32))
    cs = np.stack([np.cos(th), np.sin(th)], -1).astype(np.float32)
    lens = np.full(nseq, L, np.int32)
    lens[1] = L - 37
    if nseq > 4:
        lens[4] = 130
    bufs = []

    def up(a):
        d = ctx.alloc(a.nbytes); d.upload(a); bufs.append(d); return d
    dx, dcs, dl = up(x), up(cs), up(lens)
    dqkv = ctx.alloc(rows * 768 * 4); dctx = ctx.alloc(rows * 256 * 4); bufs += [dqkv, dctx]
    rot = C.c_int32(-1)
    ctx._chk(capi.lib.rfe_k_lightglue_self_attention(ctx.h, layer, dx.ptr, dcs.ptr, dl.ptr, nseq, L, dqkv.ptr, dctx.ptr, C.byref(rot)))
    qkv = dqkv.download((rows, 768), np.float32)
    got = dctx.download((rows, 256), np.float32)
    for d in bufs:
        d.free()
    if expect_rotated is not None:
        assert rot.value == expect_rotated
    plain, rotated, want = _f64(x, W, b, cs, lens, L)
    ref = rotated if rot.value == 1 else plain
    assert np.abs(qkv[:, 512:] - ref[:, 512:]).max() < QKV_TOL                       # v: never rotated
    assert np.abs(qkv[:, :512] - ref[:, :512]).max() < QKV_TOL, f"q | k (rotated = {rot.value})"
    for sq, n in enumerate(lens):
        assert np.abs(got[sq * L:sq * L + n] - want[sq * L:sq * L + n]).max() < CTX_TOL, f"sequence {sq}"
        assert not got[sq * L + n:(sq + 1) * L].any()                                # padded query rows: zero context
