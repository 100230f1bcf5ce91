"""GPU parity of the THROUGHPUT tiling against the oracle.

The kernels the benchmark times -- 128x256 k-permuted GEMM tiles, the LayerNorm + GELU fusion around ffn.0 / ffn.3, the one-workgroup
attention, the per-frame layer-0 self block of the stream mode -- are only selected from 32 768 token rows upwards
(launch_gemm_nt, gemm.hip), i.e. P >= 16 pairs at K = 1024.  These tests run such batches against oracle.lightglue PER PAIR on inputs
with hundreds of matches, and read the final token states / the log-assignment matrix of one pair of the batched / stream call
through the one-shot tap (rfe_k_set_lightglue_tap).  Semantics held: src/Matchers/lightglue_onnx.cpp:437-453 (the match list)."""
import numpy as np
import pytest

from rover_slam_amd import weights as Wt, synth
from tolerances import LG_SCORE_TOL, LG_STATE_TOL, LG_LOGSCORE_RTOL, LG_LOGSCORE_ATOL, lists_agree_borderline

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from rover_slam_amd import capi
    c = capi.Context(0)
    c.set_weights(capi.KIND_SUPERPOINT, Wt.make_superpoint(seed=7))
    c.set_weights(capi.KIND_LIGHTGLUE, Wt.make_lightglue(seed=11))
    yield c
    c.close()


def _constructed_batch(P, K, seed, lens0, lens1):
    """set 1 = permuted copy of a random unit-vector set 0 with 1 % descriptor noise and 0.02 keypoint noise (the construction of
    tools/lg_tolerance_study.py): several hundred true matches per pair."""
    rng = np.random.default_rng(seed)
    k0 = np.zeros((P, K, 2), np.float32); k1 = np.zeros((P, K, 2), np.float32)
    d0 = np.zeros((P, K, 256), np.float32); d1 = np.zeros((P, K, 256), np.float32)
    for p in range(P):
        a = rng.standard_normal((K, 256)).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
        kk = rng.uniform(-0.9, 0.9, (K, 2)).astype(np.float32)
        perm = rng.permutation(K)
        m, n = lens0[p], lens1[p]
        d0[p, :m] = a[:m]; k0[p, :m] = kk[:m]
        bb = a[perm][:n] + 0.01 * rng.standard_normal((n, 256)).astype(np.float32)
        d1[p, :n] = bb / np.linalg.norm(bb, axis=1, keepdims=True)
        k1[p, :n] = (kk[perm][:n] + 0.02 * rng.standard_normal((n, 2))).astype(np.float32)
    return k0, k1, d0, d1


@pytest.mark.parametrize("fold", [1, 0])
def test_lightglue_batch16_k1024_vs_oracle(ctx, oracle, fold):
    """P = 16 pairs at K = 1024 in ONE call = 32 768 token rows: qkv / cross-qkv / ffn.0 / ffn.3 take the 128x256 k-permuted tiles with
    the fused LayerNorm + GELU, the attention one workgroup per query block -- the exact instantiations of the bench step.  Every pair
    against the oracle: match list (borderline rule), match scores, > 400 matches; pair 9's final token states and log-assignment
    matrix through the tap."""
    from rover_slam_amd import capi
    P, K, tap_pair = 16, 1024, 9
    lens0 = [1024] * P; lens1 = [1024] * P
    lens0[3], lens1[5], lens0[12], lens1[12] = 700, 900, 611, 1001          # a few ragged pairs inside the full-size batch
    k0, k1, d0, d1 = _constructed_batch(P, K, 77, lens0, lens1)
    dx0, dx1, dsc = ctx.alloc(K * 1024), ctx.alloc(K * 1024), ctx.alloc(K * K * 4)
    ctx.set_option(capi.OPT_LG_FOLD_WO, fold)
    try:
        ctx._chk(capi.lib.rfe_k_set_lightglue_tap(ctx.h, tap_pair, dx0.ptr, dx1.ptr, dsc.ptr))
        S, pairs, ms = ctx.match(k0, k1, d0, d1, lens0, lens1)
    finally:
        ctx.set_option(capi.OPT_LG_FOLD_WO, 1)
    x0, x1, sc = dx0.download((K, 256), np.float32), dx1.download((K, 256), np.float32), dsc.download((K, K), np.float32)
    w = Wt.make_lightglue(seed=11)
    worst, total = 0.0, 0
    for p in range(P):
        m, n = lens0[p], lens1[p]
        r = oracle.lightglue(w, k0[p, :m], k1[p, :n], d0[p, :m], d1[p, :n], debug=True)
        ok, dev, one_sided = lists_agree_borderline(pairs[p, :S[p]], ms[p, :S[p]], r["pairs"], r["ms"], r["scores"], K)
        assert ok and dev < LG_SCORE_TOL, (p, ok, dev, one_sided)
        assert S[p] > 400 and r["S"] > 400, (p, S[p], r["S"])
        worst, total = max(worst, dev), total + int(S[p])
        if p == tap_pair:
            assert np.abs(x0[:m] - r["x0"]).max() < LG_STATE_TOL and np.abs(x1[:n] - r["x1"]).max() < LG_STATE_TOL
            assert np.abs(sc[:m, :n] - r["scores"]).max() < LG_LOGSCORE_RTOL * np.abs(r["scores"]).max()
    print(f"batch16 fold={fold}: {total} matches over {P} pairs, max |score dev| {worst:.2e}")
    for b in (dx0, dx1, dsc):
        b.free()


def test_lightglue_batch16_tap_equals_single_pair_tap(ctx):
    """The tap itself: pair 2 of a P = 16 batch (throughput tiles) against the same pair alone through rfe_k_lightglue_taps
    (64-row latency tiles, split-key attention): token states within the stated tolerance of each other."""
    from rover_slam_amd import capi
    P, K = 16, 1024
    k0, k1, d0, d1 = _constructed_batch(P, K, 5, [1024] * P, [1024] * P)
    dx0, dx1 = ctx.alloc(K * 1024), ctx.alloc(K * 1024)
    ctx._chk(capi.lib.rfe_k_set_lightglue_tap(ctx.h, 2, dx0.ptr, dx1.ptr, None))
    ctx.match(k0, k1, d0, d1, [K] * P, [K] * P)
    xb0, xb1 = dx0.download((K, 256), np.float32), dx1.download((K, 256), np.float32)
    bufs = [ctx.alloc(a.nbytes).upload(a) for a in (k0[2], k1[2], d0[2], d1[2])]
    ctx._chk(capi.lib.rfe_k_lightglue_taps(ctx.h, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, K, K, dx0.ptr, dx1.ptr, None))
    xs0, xs1 = dx0.download((K, 256), np.float32), dx1.download((K, 256), np.float32)
    assert np.abs(xb0 - xs0).max() < LG_STATE_TOL and np.abs(xb1 - xs1).max() < LG_STATE_TOL
    # a tap is one-shot: the next call leaves the buffers alone
    dx0.upload(np.zeros((K, 256), np.float32))
    ctx.match(k0[:1], k1[:1], d0[:1], d1[:1], [K], [K])
    assert not dx0.download((K, 256), np.float32).any()
    for b in bufs + [dx0, dx1]:
        b.free()


@pytest.mark.parametrize("fp16x2", [0, 1])
def test_stream_b33_filter0_vs_oracle(ctx, oracle, fp16x2):
    """(fp16x2 = 1: the same stream with RFE_OPT_LG_FP16X2 on -- split Linears and attention, same bars.)
    configs[3]'s per-GPU shard through rfe_extract_match_stream_dev (B = 33 frames 640x480, Kmax = 1024: per-frame layer-0 self
    block, throughput tiles) with filter_thr = 0.0, so EVERY mutual pair is emitted (with random weights only a handful pass 0.1).
    Frames 9, 17 and 25 repeat their predecessor: those pairs give 200+ mutual matches, the others of the bench stream the 50-100
    mutual nearest neighbours random weights leave.  Six pairs against the oracle (borderline rule, scores); pair 16
    (a repeated frame) also with its final token states and log-assignment matrix through the tap."""
    from rover_slam_amd import capi
    B, H, W, K, tap_pair = 33, 480, 640, 1024, 16
    frames, _ = synth.make_frames(B, H, W, seed=20240314)
    dup = (8, 16, 24)
    for i in dup:
        frames[i + 1] = frames[i]
    dimg = ctx.alloc(frames.nbytes).upload(frames)
    dn, dk, ds, dd = ctx.alloc(B * 4), ctx.alloc(B * K * 8), ctx.alloc(B * K * 4), ctx.alloc(B * K * 1024)
    dS, dp, dm = ctx.alloc((B - 1) * 4), ctx.alloc((B - 1) * K * 8), ctx.alloc((B - 1) * K * 4)
    dx0, dx1, dsc = ctx.alloc(K * 1024), ctx.alloc(K * 1024), ctx.alloc(K * K * 4)
    ctx.set_option(capi.OPT_LG_FP16X2, fp16x2)
    try:
        ctx._chk(capi.lib.rfe_k_set_lightglue_tap(ctx.h, tap_pair, dx0.ptr, dx1.ptr, dsc.ptr))
        ctx._chk(capi.lib.rfe_extract_match_stream_dev(ctx.h, dimg.ptr, H, W, W, B, K, 0.0005, 0.0, dn.ptr, dk.ptr, ds.ptr, dd.ptr,
                                                       dS.ptr, dp.ptr, dm.ptr))
        ctx.synchronize()
    finally:
        ctx.set_option(capi.OPT_LG_FP16X2, 0)
    n, kxy, desc = dn.download((B,), np.int32), dk.download((B, K, 2), np.int32), dd.download((B, K, 256), np.float32)
    S, pairs, ms = dS.download((B - 1,), np.int32), dp.download((B - 1, K, 2), np.int32), dm.download((B - 1, K), np.float32)
    x0, x1, sc = dx0.download((K, 256), np.float32), dx1.download((K, 256), np.float32), dsc.download((K, K), np.float32)
    wlg = Wt.make_lightglue(seed=11)
    total, worst = 0, 0.0
    for i in (0, 7, 8, tap_pair, 24, 31):
        kn = [oracle.normalize_keypoints(kxy[j, :n[j]].astype(np.float32), H, W) for j in (i, i + 1)]   # extraction is bit-exact (other tests)
        r = oracle.lightglue(wlg, kn[0], kn[1], desc[i, :n[i]], desc[i + 1, :n[i + 1]], filter_thr=0.0, debug=True)
        ok, dev, one_sided = lists_agree_borderline(pairs[i, :S[i]], ms[i, :S[i]], r["pairs"], r["ms"], r["scores"], K, filter_thr=0.0)
        assert ok and dev < LG_SCORE_TOL, (i, ok, dev, one_sided)
        assert S[i] > (150 if i in dup else 20), (i, S[i])     # measured: 216 / 59 (random weights)
        total, worst = total + int(S[i]), max(worst, dev)
        if i == tap_pair:
            assert np.abs(x0[:n[i]] - r["x0"]).max() < LG_STATE_TOL and np.abs(x1[:n[i + 1]] - r["x1"]).max() < LG_STATE_TOL
            dlog = np.abs(sc[:n[i], :n[i + 1]] - r["scores"]).max()
            dprob = np.abs(np.exp(sc[:n[i], :n[i + 1]]) - np.exp(r["scores"])).max()
            print(f"tap pair {i}: max |log-score dev| {dlog:.2e} (max |log-score| {np.abs(r['scores']).max():.1f}), max |probability dev| {dprob:.2e}")
            assert dprob < LG_SCORE_TOL and dlog < LG_LOGSCORE_ATOL, (dlog, dprob)
    print(f"stream B=33 filter 0 (fp16x2 = {fp16x2}): {total} matches over 6 pairs, max |score dev| {worst:.2e}, S = {S.tolist()}")
    for d in (dimg, dn, dk, ds, dd, dS, dp, dm, dx0, dx1, dsc):
        d.free()


def test_fused_layernorm_rows_with_large_mean(ctx):
    """the LayerNorm statistics of the fused ffn.0 -> LN -> GELU -> ffn.3 path come from per-tile partials; a one-pass
    E[x^2] - mean^2 would cancel when |mean| >> std, which seeded synthetic weights never produce but real ones may.  Here ffn.0's
    bias is raised by 1000 (rows of mean ~1000, std ~1): the fused throughput path (32 768 rows), the stand-alone lg_ln_gelu
    path (the same rows in chunks of 4096: 64-row tiles) and a float64 evaluation must agree."""
    from rover_slam_amd import capi
    from scipy.special import erf
    w = Wt.make_lightglue(seed=11)
    man, _ = Wt.lg_manifest()
    t = {name: (off, shape) for name, off, shape in man}
    off, shape = t["layers.0.self.b1"]
    w2 = w.copy()
    w2[off:off + 512] += np.float32(1000.0)
    c2 = capi.Context(0)
    c2.set_weights(capi.KIND_LIGHTGLUE, w2)
    rows = 32768
    rng = np.random.default_rng(8)
    x = rng.standard_normal((rows, 256)).astype(np.float32)
    s = rng.standard_normal((rows, 256)).astype(np.float32)
    dx, dsec, dout = c2.alloc(x.nbytes).upload(x), c2.alloc(s.nbytes).upload(s), c2.alloc(x.nbytes)
    c2._chk(capi.lib.rfe_k_lightglue_ffn(c2.h, 0, 0, dx.ptr, dsec.ptr, rows, dout.ptr))
    fused = dout.download((rows, 256), np.float32)
    chunk = 4096
    alone = np.empty_like(fused)
    for r0 in range(0, rows, chunk):
        c2._chk(capi.lib.rfe_k_lightglue_ffn(c2.h, 0, 0, dx.ptr + r0 * 1024, dsec.ptr + r0 * 1024, chunk, dout.ptr))
        alone[r0:r0 + chunk] = dout.download((chunk, 256), np.float32)
    g = lambda name: w2[t[name][0]:t[name][0] + int(np.prod(t[name][1]))].reshape(t[name][1]).astype(np.float64)
    sel = rng.choice(rows, 512, replace=False)
    h = np.concatenate([x[sel], s[sel]], 1).astype(np.float64) @ g("layers.0.self.W1").T + g("layers.0.self.b1")
    assert h.mean() > 900 and h.std(axis=1).mean() < 5
    mu, var = h.mean(1, keepdims=True), h.var(1, keepdims=True)
    hn = (h - mu) / np.sqrt(var + 1e-5) * g("layers.0.self.ln_g") + g("layers.0.self.ln_b")
    ref = x[sel] + (0.5 * hn * (1 + erf(hn / np.sqrt(2.0)))) @ g("layers.0.self.W2").T + g("layers.0.self.b2")
    d_f, d_a, d_fa = np.abs(fused[sel] - ref).max(), np.abs(alone[sel] - ref).max(), np.abs(fused - alone).max()
    print(f"large-mean rows: fused vs f64 {d_f:.2e}, stand-alone vs f64 {d_a:.2e}, fused vs stand-alone {d_fa:.2e}")
    assert d_f < 3e-3 and d_a < 3e-3 and d_fa < 3e-3, (d_f, d_a, d_fa)
    for b in (dx, dsec, dout):
        b.free()
    c2.close()


def test_lightglue_batch16_fp16x2_vs_oracle(ctx, oracle):
    """RFE_OPT_LG_FP16X2 (default off): the Linears and the attention of a 16-pair call as split products on the f16 matrix pipe (gemm_h2.hip,
    lg_attention_h2.hip: fp16 hi + lo operands, three products, fp32 accumulation).  Same bar as the fp32 path: every pair against the oracle -- match list (borderline
    rule), match scores within LG_SCORE_TOL, > 400 matches -- and one pair's final token states / log-assignment matrix through the tap.
    Also: the option really changes the arithmetic (results differ from the fp32 path in the last bits) and switches back."""
    from rover_slam_amd import capi
    P, K, tap_pair = 16, 1024, 5
    lens0 = [1024] * P; lens1 = [1024] * P
    lens0[2], lens1[7], lens0[11], lens1[11] = 650, 980, 777, 1003
    k0, k1, d0, d1 = _constructed_batch(P, K, 91, lens0, lens1)
    dx0, dx1, dsc = ctx.alloc(K * 1024), ctx.alloc(K * 1024), ctx.alloc(K * K * 4)
    assert ctx.get_option(capi.OPT_LG_FP16X2) == 0
    S32, pairs32, ms32 = ctx.match(k0, k1, d0, d1, lens0, lens1)
    ctx.set_option(capi.OPT_LG_FP16X2, 1)
    try:
        assert ctx.get_option(capi.OPT_LG_FP16X2) == 1
        ctx._chk(capi.lib.rfe_k_set_lightglue_tap(ctx.h, tap_pair, dx0.ptr, dx1.ptr, dsc.ptr))
        S, pairs, ms = ctx.match(k0, k1, d0, d1, lens0, lens1)
    finally:
        ctx.set_option(capi.OPT_LG_FP16X2, 0)
    total, worst = 0, 0.0
    for p in range(P):
        m, n = lens0[p], lens1[p]
        ref = oracle.lightglue(Wt.make_lightglue(seed=11), k0[p, :m], k1[p, :n], d0[p, :m], d1[p, :n], debug=True)
        ok, dev, only = lists_agree_borderline(pairs[p, :S[p]], ms[p, :S[p]], ref["pairs"], ref["ms"], ref["scores"], K)
        assert ok and dev < LG_SCORE_TOL, (p, int(S[p]), ref["S"], dev, only)
        assert ref["S"] > 400
        total += int(S[p]); worst = max(worst, dev)
        if p == tap_pair:
            x0 = dx0.download((K, 256), np.float32)[:m]; x1 = dx1.download((K, 256), np.float32)[:n]
            assert np.abs(x0 - ref["x0"]).max() < LG_STATE_TOL and np.abs(x1 - ref["x1"]).max() < LG_STATE_TOL
            sc = dsc.download((K, K), np.float32)[:m, :n]
            assert np.abs(np.exp(sc) - np.exp(ref["scores"])).max() < LG_SCORE_TOL
    differs = any(not np.array_equal(ms[p, :S[p]], ms32[p, :S32[p]]) for p in range(P) if S[p] == S32[p])
    assert differs, "RFE_OPT_LG_FP16X2 = 1 produced bit-identical scores: the split GEMM did not run"
    S_b, pairs_b, ms_b = ctx.match(k0, k1, d0, d1, lens0, lens1)          # option off again: the fp32 path, bit for bit
    assert np.array_equal(S_b, S32) and all(np.array_equal(ms_b[p, :S32[p]], ms32[p, :S32[p]]) for p in range(P))
    print(f"batch16 fp16x2: {total} matches over {P} pairs, max |score dev| vs oracle {worst:.2e}")
    for b in (dx0, dx1, dsc):
        b.free()


def test_lightglue_one_pair_fp16x2_vs_oracle(ctx, oracle):
    """RFE_OPT_LG_FP16X2 at the reference's own shape, ONE pair per call: the Linears run the split form of the latency tiles (gemm_lat.hip, H2: the
    same ring / tile / epilogue, three v_mfma_f32_16x16x32_f16 per 32 k), the attention runs the split form of lg_attention_lat.hip (launch_lg_attention passes the option on, as rover_fe.h says).  Same bar as the fp32
    path (match list by the borderline rule, scores within LG_SCORE_TOL, final token states), the option must really change the arithmetic, and a
    ragged pair must work as well."""
    from rover_slam_amd import capi
    K = 1024
    for lens0, lens1, seed in (([1024], [1024], 93), ([701], [1003], 94)):
        k0, k1, d0, d1 = _constructed_batch(1, K, seed, lens0, lens1)
        dx0, dx1, dsc = ctx.alloc(K * 1024), ctx.alloc(K * 1024), ctx.alloc(K * K * 4)
        S32, pairs32, ms32 = ctx.match(k0, k1, d0, d1, lens0, lens1)
        ctx.set_option(capi.OPT_LG_FP16X2, 1)
        try:
            ctx._chk(capi.lib.rfe_k_set_lightglue_tap(ctx.h, 0, dx0.ptr, dx1.ptr, dsc.ptr))
            S, pairs, ms = ctx.match(k0, k1, d0, d1, lens0, lens1)
        finally:
            ctx.set_option(capi.OPT_LG_FP16X2, 0)
        m, n = lens0[0], lens1[0]
        ref = oracle.lightglue(Wt.make_lightglue(seed=11), k0[0, :m], k1[0, :n], d0[0, :m], d1[0, :n], debug=True)
        ok, dev, only = lists_agree_borderline(pairs[0, :S[0]], ms[0, :S[0]], ref["pairs"], ref["ms"], ref["scores"], K)
        assert ok and dev < LG_SCORE_TOL and ref["S"] > 400, (int(S[0]), ref["S"], dev, only)
        x0 = dx0.download((K, 256), np.float32)[:m]; x1 = dx1.download((K, 256), np.float32)[:n]
        assert np.abs(x0 - ref["x0"]).max() < LG_STATE_TOL and np.abs(x1 - ref["x1"]).max() < LG_STATE_TOL
        assert S[0] != S32[0] or not np.array_equal(ms[0, :S[0]], ms32[0, :S32[0]]), "the option did not change the arithmetic: the split tiles did not run"
        print(f"one pair ({m} x {n}) fp16x2: {int(S[0])} matches, max |score dev| vs oracle {dev:.2e}")
        for b in (dx0, dx1, dsc):
            b.free()


@pytest.mark.parametrize("rows", [34077, 2013])
def test_ffn_block_ragged_rows_fp32_and_fp16x2_vs_float64(ctx, rows):
    """One FFN block (ffn.0 -> LayerNorm -> GELU -> ffn.3 + residual) on 34 077 token rows -- not a multiple of the 128-row tiles -- through the
    forward's own code (rfe_k_lightglue_ffn: throughput tiles, fused LayerNorm partials) with the fp32 kernels and with RFE_OPT_LG_FP16X2 (split
    GEMMs: weight planes by LDS-DMA, clamped edge rows): both against a float64 evaluation on sampled rows including the last ones.
    2 013 rows: the one-pair latency tiles (gemm_lat.hip: 16 x 16 x 4 fp32 / 16 x 16 x 32 split-fp16 products behind the LDS-DMA ring, stand-alone
    LayerNorm + GELU pass, residual epilogue), also not a multiple of the 64-row tile."""
    from rover_slam_amd import capi
    from scipy.special import erf
    w = Wt.make_lightglue(seed=11)
    man, _ = Wt.lg_manifest()
    t = {name: (off, shape) for name, off, shape in man}
    g = lambda name: w[t[name][0]:t[name][0] + int(np.prod(t[name][1]))].reshape(t[name][1]).astype(np.float64)
    rng = np.random.default_rng(21)
    x = rng.standard_normal((rows, 256)).astype(np.float32)
    s = rng.standard_normal((rows, 256)).astype(np.float32)
    dx, dsec, dout = ctx.alloc(x.nbytes).upload(x), ctx.alloc(s.nbytes).upload(s), ctx.alloc(x.nbytes)
    sel = np.concatenate([rng.choice(rows - 200, 300, replace=False), np.arange(rows - 200, rows)])
    p = "layers.3.cross."
    h = np.concatenate([x[sel], s[sel]], 1).astype(np.float64) @ g(p + "W1").T + g(p + "b1")
    mu, var = h.mean(1, keepdims=True), h.var(1, keepdims=True)
    hn = (h - mu) / np.sqrt(var + 1e-5) * g(p + "ln_g") + g(p + "ln_b")
    ref = x[sel] + (0.5 * hn * (1 + erf(hn / np.sqrt(2.0)))) @ g(p + "W2").T + g(p + "b2")
    dev = {}
    for opt in (0, 1):
        ctx.set_option(capi.OPT_LG_FP16X2, opt)
        try:
            ctx._chk(capi.lib.rfe_k_lightglue_ffn(ctx.h, 3, 1, dx.ptr, dsec.ptr, rows, dout.ptr))
        finally:
            ctx.set_option(capi.OPT_LG_FP16X2, 0)
        out = dout.download((rows, 256), np.float32)
        assert np.isfinite(out).all()
        dev[opt] = float(np.abs(out[sel] - ref).max())
    print(f"FFN block, {rows} rows: max |out - float64|  fp32 kernels {dev[0]:.2e}   fp16x2 split kernels {dev[1]:.2e}  (|out| up to {np.abs(ref).max():.1f})")
    assert dev[0] < 1e-4 and dev[1] < 1e-4, dev
    for b in (dx, dsec, dout):
        b.free()
