"""north_star's 1e-4 where trained weights live, and the big-batch shards of configs[3] (item 5).

The seeded default LightGlue law is ill-conditioned (log-assignment up to |850|: one fp32 ulp = 6e-5 in the log domain, any two fp32
evaluations differ by 1-3e-4, tests/tolerances.py) -- its 5e-4 bar says nothing about the kernels.  `weights.make_lightglue(calibrated=True)`
keeps token norms O(1) and the log-assignment at 52-63; on it the one-pair (latency tiles), 16-pair (throughput tiles) and B = 33 stream paths
are held to 1e-4 with IDENTICAL match lists, against oracle.lightglue per pair.  Semantics: src/Matchers/lightglue_onnx.cpp:437-453."""
import numpy as np
import pytest

from rover_slam_amd import weights as Wt, synth
from tolerances import LG_SCORE_TOL_CALIBRATED, LG_STATE_TOL
from test_gpu_throughput_parity import _constructed_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wcal():
    return Wt.make_superpoint(seed=7, desc_center="auto"), Wt.make_lightglue(seed=11, calibrated=True)


@pytest.fixture(scope="module")
def ctx(wcal):
    from rover_slam_amd import capi
    c = capi.Context(0)
    c.set_weights(capi.KIND_SUPERPOINT, wcal[0])
    c.set_weights(capi.KIND_LIGHTGLUE, wcal[1])
    yield c
    c.close()


def _identical(pairs, ms, r):
    same = len(pairs) == r["S"] and np.array_equal(pairs, r["pairs"])
    return same, (float(np.abs(ms - r["ms"]).max()) if same and len(ms) else float("nan"))


@pytest.mark.parametrize("fold", [1, 0])
def test_one_pair_k1024_calibrated_1e4(ctx, oracle, wcal, fold):
    """ONE pair per call (the reference's own shape, lightglue_onnx.cpp:168-172): latency tiles, full and ragged"""
    from rover_slam_amd import capi
    K = 1024
    ctx.set_option(capi.OPT_LG_FOLD_WO, fold)
    try:
        worst, total = 0.0, 0
        for lens0, lens1, seed in (([1024], [1024], 93), ([701], [1003], 94), ([1024], [640], 95)):
            k0, k1, d0, d1 = _constructed_batch(1, K, seed, lens0, lens1)
            S, pairs, ms = ctx.match(k0, k1, d0, d1, lens0, lens1)
            r = oracle.lightglue(wcal[1], k0[0, :lens0[0]], k1[0, :lens1[0]], d0[0, :lens0[0]], d1[0, :lens1[0]], debug=True)
            same, dev = _identical(pairs[0, :S[0]], ms[0, :S[0]], r)
            assert same and dev < LG_SCORE_TOL_CALIBRATED, (seed, same, dev)
            assert r["S"] > 0.6 * min(lens0[0], lens1[0]) and np.abs(r["scores"]).max() < 70.0
            worst, total = max(worst, dev), total + r["S"]
        print(f"one pair, calibrated weights, fold = {fold}: {total} matches, lists identical, max |score dev| {worst:.2e}")
    finally:
        ctx.set_option(capi.OPT_LG_FOLD_WO, 1)


@pytest.mark.parametrize("K", [48, 160, 256])
def test_one_pair_small_k_calibrated_1e4(ctx, oracle, wcal, K):
    """K <= 256 was held to 1e-4 until the latency tiling stopped summing k in the oracle's order (tests/tolerances.py LG_SCORE_TOL_SMALL = 2e-4 on the
    ill-conditioned seeded set).  On the calibrated set the 1e-4 bar holds at these sizes too, lists identical -- later drift stays visible here."""
    k0, k1, d0, d1 = _constructed_batch(1, K, 500 + K, [K], [K - K // 5])
    S, pairs, ms = ctx.match(k0, k1, d0, d1, [K], [K - K // 5])
    r = oracle.lightglue(wcal[1], k0[0], k1[0, :K - K // 5], d0[0], d1[0, :K - K // 5])
    same, dev = _identical(pairs[0, :S[0]], ms[0, :S[0]], r)
    assert same and dev < LG_SCORE_TOL_CALIBRATED and r["S"] > K // 3, (K, same, dev, r["S"])
    print(f"one pair K = {K}, calibrated weights: {r['S']} matches, lists identical, max |score dev| {dev:.2e}")


def test_batch16_k1024_calibrated_1e4(ctx, oracle, wcal):
    """16 pairs per call = 32 768 token rows: the THROUGHPUT tiling the benchmark times, ragged lengths included; final token states and
    log-assignment of one pair through the tap"""
    from rover_slam_amd import capi
    P, K, tap = 16, 1024, 5
    rng = np.random.default_rng(7)
    lens0 = [1024] * 8 + [int(v) for v in rng.integers(500, 1025, 8)]
    lens1 = [1024] * 8 + [int(v) for v in rng.integers(500, 1025, 8)]
    k0, k1, d0, d1 = _constructed_batch(P, K, 321, lens0, lens1)
    dx0, dx1, dsc = ctx.alloc(K * 1024), ctx.alloc(K * 1024), ctx.alloc(K * K * 4)
    ctx._chk(capi.lib.rfe_k_set_lightglue_tap(ctx.h, tap, dx0.ptr, dx1.ptr, dsc.ptr))
    S, pairs, ms = ctx.match(k0, k1, d0, d1, lens0, lens1)
    x0, x1, sc = dx0.download((K, 256), np.float32), dx1.download((K, 256), np.float32), dsc.download((K, K), np.float32)
    worst, total = 0.0, 0
    for p in range(P):
        r = oracle.lightglue(wcal[1], k0[p, :lens0[p]], k1[p, :lens1[p]], d0[p, :lens0[p]], d1[p, :lens1[p]], debug=True)
        same, dev = _identical(pairs[p, :S[p]], ms[p, :S[p]], r)
        assert same and dev < LG_SCORE_TOL_CALIBRATED, (p, same, dev)
        worst, total = max(worst, dev), total + r["S"]
        if p == tap:
            assert np.abs(x0[:lens0[p]] - r["x0"]).max() < LG_STATE_TOL and np.abs(x1[:lens1[p]] - r["x1"]).max() < LG_STATE_TOL
            dlog = np.abs(sc[:lens0[p], :lens1[p]] - r["scores"]).max()
            assert dlog < 2e-4, dlog             # the log-assignment matrix itself (|values| <= 63): a few fp32 ulps
            print(f"tap pair {p}: max |log-score dev| {dlog:.2e} at max |log-score| {np.abs(r['scores']).max():.1f}")
    assert total > 5000
    print(f"16 pairs, calibrated weights: {total} matches, lists identical, max |score dev| {worst:.2e}")
    for d in (dx0, dx1, dsc):
        d.free()


@pytest.mark.parametrize("tag", ["e", "f"])
def test_one_pair_vs_hf_fixture_calibrated_1e4(ctx, golden_dir, tag):
    """the HIP path against HuggingFace transformers' LightGlue on the calibrated law (tests/golden/lg_e.npz, lg_f.npz; tools/gen_golden.py calibrated):
    an implementation that shares no code with this repo -- identical match lists, scores within 1e-4"""
    import gen_golden as G
    g = np.load(f"{golden_dir}/lg_{tag}.npz")
    k0, k1, d0, d1, _ = G.calibrated_case(tag)
    S, pairs, ms = ctx.match(k0[None], k1[None], d0[None], d1[None], [len(k0)], [len(k1)])
    assert S[0] == len(g["pairs"]) and np.array_equal(pairs[0, :S[0]], g["pairs"])
    dev = float(np.abs(ms[0, :S[0]] - g["ms"]).max())
    assert dev < LG_SCORE_TOL_CALIBRATED, dev
    print(f"HIP vs HF (calibrated law, lg_{tag}): {S[0]} matches, lists identical, max |score dev| {dev:.2e}")


def _run_stream(ctx, frames, K, filter_thr=0.1):
    from rover_slam_amd import capi
    B, H, W = frames.shape
    dimg = ctx.alloc(frames.nbytes).upload(frames)
    dn, dk, ds, dd = ctx.alloc(B * 4), ctx.alloc(B * K * 8), ctx.alloc(B * K * 4), ctx.alloc(B * K * 1024)
    dS, dp, dm = ctx.alloc((B - 1) * 4), ctx.alloc((B - 1) * K * 8), ctx.alloc((B - 1) * K * 4)
    ctx._chk(capi.lib.rfe_extract_match_stream_dev(ctx.h, dimg.ptr, H, W, W, B, K, 0.0005, filter_thr, dn.ptr, dk.ptr, ds.ptr, dd.ptr, dS.ptr, dp.ptr, dm.ptr))
    ctx.synchronize()
    out = dict(n=dn.download((B,), np.int32), kxy=dk.download((B, K, 2), np.int32), score=ds.download((B, K), np.float32),
               S=dS.download((B - 1,), np.int32), pairs=dp.download((B - 1, K, 2), np.int32), ms=dm.download((B - 1, K), np.float32))
    out["_bufs"] = (dimg, dn, dk, ds, dd, dS, dp, dm)
    return out


def _check_pairs(oracle, wcal, frames, out, which, K, desc_of):
    H, W = frames.shape[1:]
    total, worst = 0, 0.0
    for i in which:
        a, b = (oracle.superpoint(wcal[0], frames[j], kmax=K) for j in (i, i + 1))
        for j, r in ((i, a), (i + 1, b)):                                   # extraction: bit-exact, descriptors included
            assert out["n"][j] == r["n"] and np.array_equal(out["kxy"][j], r["kxy"]) and np.array_equal(out["score"][j], r["score"]), (i, j)
            assert np.array_equal(desc_of(j), r["desc"]), (i, j)
        lg = oracle.lightglue(wcal[1], oracle.normalize_keypoints(a["kxy"][:a["n"]].astype(np.float32), H, W),
                              oracle.normalize_keypoints(b["kxy"][:b["n"]].astype(np.float32), H, W), a["desc"][:a["n"]], b["desc"][:b["n"]])
        S = out["S"][i]
        same, dev = _identical(out["pairs"][i, :S], out["ms"][i, :S], lg)
        assert same and dev < LG_SCORE_TOL_CALIBRATED, (i, same, dev, S, lg["S"])
        total, worst = total + int(S), max(worst, dev)
    return total, worst


def test_stream_b33_calibrated_1e4(ctx, oracle, wcal):
    """the bench's own batch (33 frames 640 x 480, cell-aligned shifts, Kmax 1024, filter 0.1): six pairs against the oracle, identical lists, 1e-4"""
    B, K = 33, 1024
    frames, _ = synth.make_frames(B, 480, 640, seed=20240314, max_shift=16, shift_step=8)
    out = _run_stream(ctx, frames, K)
    desc = out["_bufs"][4].download((B, K, 256), np.float32)
    total, worst = _check_pairs(oracle, wcal, frames, out, (0, 7, 15, 16, 24, 31), K, lambda j: desc[j])
    assert total > 600 and out["S"].min() > 50, (total, out["S"].tolist())
    print(f"stream B = 33, calibrated pairing: {total} matches over 6 pairs (S per pair {out['S'].tolist()}), lists identical, max |score dev| {worst:.2e}")
    for d in out["_bufs"]:
        d.free()


@pytest.mark.parametrize("B", [129, 257])
def test_stream_big_batches_of_configs3_strong_scaling(oracle, wcal, B):
    """configs[3] under strong scaling puts 257 (N = 1) or 129 (N = 2) frames into ONE rfe_extract_match_stream_dev call (bench.py --scaling strong):
    first / middle / last pair against the oracle (extraction bit-exact, match lists identical, 1e-4), every frame saturating Kmax, the first 33
    frames equal to the B = 33 call's, and the workspace high-water mark reported.  Own ctx: the workspace of this call is ~100 GB."""
    from rover_slam_amd import capi
    K = 1024
    frames, _ = synth.make_frames(B, 480, 640, seed=20240314, max_shift=16, shift_step=8)
    c = capi.Context(0)
    try:
        c.set_weights(capi.KIND_SUPERPOINT, wcal[0]); c.set_weights(capi.KIND_LIGHTGLUE, wcal[1])
        out = _run_stream(c, frames, K)
        ws = c.workspace_bytes()
        dd = out["_bufs"][4]
        rows = {}

        def desc_of(j):                                      # one frame's descriptors (the whole array is B MB)
            if j not in rows:
                a = np.empty((K, 256), np.float32)
                c._chk(capi.lib.rfe_memcpy_d2h(c.h, a.ctypes.data, dd.ptr + j * K * 1024, K * 1024))
                rows[j] = a
            return rows[j]
        assert out["n"].min() == K and out["S"].min() > 50, (out["n"].min(), out["S"].min())
        which = (0, (B - 1) // 2, B - 2)
        total, worst = _check_pairs(oracle, wcal, frames, out, which, K, desc_of)
        small = _run_stream(c, frames[:33], K)
        assert np.array_equal(small["kxy"], out["kxy"][:33]) and np.array_equal(small["S"], out["S"][:32])
        assert all(np.array_equal(small["pairs"][q, :small["S"][q]], out["pairs"][q, :small["S"][q]]) and
                   np.array_equal(small["ms"][q, :small["S"][q]], out["ms"][q, :small["S"][q]]) for q in range(32))
        print(f"stream B = {B}: pairs {which} against the oracle: {total} matches, lists identical, max |score dev| {worst:.2e}; "
              f"workspace high-water mark {ws / 2 ** 30:.2f} GiB; matches per pair min / mean {out['S'].min()} / {out['S'].mean():.0f}")
        for d in out["_bufs"] + small["_bufs"]:
            d.free()
    finally:
        c.close()


def test_k2048_one_pair_and_three_pairs_calibrated_1e4(ctx, oracle, wcal):
    """K = 2048 keypoints per image (max_num_keypoints of the published LightGlue-style exports' other setting, tests/test_onnx_hparams.py; the reference reads K off
    the output tensor, src/Extractors/superpoint_onnx.cc:169-181, and passes whatever it gets to lightglue_sim.onnx): sequences of 2048 tokens through the
    one-pair path (4096 rows: latency tiles, 2048-key attention) and through a 3-pair call (12 288 rows: throughput tiles, the rotary epilogue of the qkv
    projection, lg_attention_dma_kernel with 16 query blocks per sequence), ragged lengths included; pairs 0 and 2 against the oracle."""
    K = 2048
    lens0, lens1 = [2048, 1800, 1531], [2048, 2048, 1999]
    k0, k1, d0, d1 = _constructed_batch(3, K, 777, lens0, lens1)
    S3, pairs3, ms3 = ctx.match(k0, k1, d0, d1, lens0, lens1)
    S1, pairs1, ms1 = ctx.match(k0[:1], k1[:1], d0[:1], d1[:1], lens0[:1], lens1[:1])
    for p in (0, 2):
        r = oracle.lightglue(wcal[1], k0[p, :lens0[p]], k1[p, :lens1[p]], d0[p, :lens0[p]], d1[p, :lens1[p]])
        same, dev = _identical(pairs3[p, :S3[p]], ms3[p, :S3[p]], r)
        assert same and dev < LG_SCORE_TOL_CALIBRATED and r["S"] > 0.5 * min(lens0[p], lens1[p]), (p, same, dev, r["S"])
        print(f"K = 2048, pair {p} of a 3-pair call: {r['S']} matches, lists identical, max |score dev| {dev:.2e}")
        if p == 0:
            same, dev = _identical(pairs1[0, :S1[0]], ms1[0, :S1[0]], r)
            assert same and dev < LG_SCORE_TOL_CALIBRATED, (same, dev)
            print(f"K = 2048, one pair per call: lists identical, max |score dev| {dev:.2e}")


def test_k4096_one_pair_calibrated_1e4(ctx, oracle, wcal):
    """the largest keypoint budget the C ABI accepts (Kmax <= 4096, include/rover_fe.h): one pair of 4096 / 3900 keypoints = 8192 token rows, the upper
    edge of the one-pair tilings; a 64 MB log-assignment matrix"""
    K = 4096
    k0, k1, d0, d1 = _constructed_batch(1, K, 4242, [4096], [3900])
    S, pairs, ms = ctx.match(k0, k1, d0, d1, [4096], [3900])
    r = oracle.lightglue(wcal[1], k0[0], k1[0, :3900], d0[0], d1[0, :3900])
    same, dev = _identical(pairs[0, :S[0]], ms[0, :S[0]], r)
    assert same and dev < LG_SCORE_TOL_CALIBRATED and r["S"] > 1500, (same, dev, r["S"])
    print(f"K = 4096, one pair: {r['S']} matches, lists identical, max |score dev| {dev:.2e}")
