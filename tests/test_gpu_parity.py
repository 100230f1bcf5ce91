"""GPU parity tests: the HIP path (through the C ABI of librover_fe.so) against the CPU oracle on the
same seeded inputs, and against the committed golden fixtures.

Bars (BASELINE.json north_star): SuperPoint keypoints / scores bit-exact after NMS, descriptors
within 1e-4 (they are in fact bit-exact: the kernels reproduce the oracle's fmaf-chain orders);
LightGlue match assignments identical, scores / token states within a stated fp32 tolerance."""
import numpy as np
import pytest

from rover_slam_amd import weights as Wt, synth
from tolerances import LG_SCORE_TOL, LG_SCORE_TOL_SMALL, LG_STATE_TOL, LG_LOGSCORE_RTOL, lists_agree

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from rover_slam_amd import capi
    c = capi.Context(0)
    c.set_weights(capi.KIND_SUPERPOINT, Wt.make_superpoint(seed=7))
    c.set_weights(capi.KIND_LIGHTGLUE, Wt.make_lightglue(seed=11))
    yield c
    c.close()


def _dev(ctx, arr):
    arr = np.ascontiguousarray(arr)
    return ctx.alloc(arr.nbytes).upload(arr)


def test_options_api(ctx):
    """Per-ctx options replace environment switches: RFE_OPT_LG_FOLD_WO defaults to 1, can be read back, unknown ids fail."""
    from rover_slam_amd import capi
    assert ctx.get_option(capi.OPT_LG_FOLD_WO) == 1
    ctx.set_option(capi.OPT_LG_FOLD_WO, 0)
    assert ctx.get_option(capi.OPT_LG_FOLD_WO) == 0
    ctx.set_option(capi.OPT_LG_FOLD_WO, 1)
    with pytest.raises(capi.RfeError, match="unknown option"):
        ctx.set_option(12345, 1)


# ------------------------------------------------------------------ kernel level
@pytest.mark.parametrize("M,K,N", [(64, 32, 64), (200, 256, 65), (4800, 256, 256), (37, 512, 130)])
def test_linear_bitexact(ctx, oracle, M, K, N):
    from rover_slam_amd import capi
    rng = np.random.default_rng(M + N)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32)
    b = rng.standard_normal((N,)).astype(np.float32)
    da, dout = _dev(ctx, a), ctx.alloc(M * N * 4)
    ctx._chk(capi.lib.rfe_k_linear(ctx.h, da.ptr, M, K, w.ctypes.data, b.ctypes.data, N, 0, dout.ptr))
    got = dout.download((M, N), np.float32)
    ref = oracle.linear(a, w, b)
    assert np.array_equal(got, ref)  # MFMA f32 == k-ordered fmaf chain
    da.free(); dout.free()


@pytest.mark.parametrize("H,W,Cin,Cout,relu,pool", [
    (16, 32, 16, 64, 1, 0), (24, 40, 64, 64, 1, 1), (60, 80, 128, 128, 1, 0), (20, 24, 64, 128, 0, 0),
    (30, 46, 32, 64, 1, 1), (60, 80, 128, 256, 1, 0),
    (28, 44, 128, 128, 1, 1), (30, 38, 128, 64, 1, 1), (120, 80, 128, 128, 1, 1), (27, 45, 128, 128, 1, 1)])     # the pooled 16 x 16 x 4 tiles (ragged edges; odd sizes fall back)
def test_conv3x3_bitexact(ctx, oracle, H, W, Cin, Cout, relu, pool):
    from rover_slam_amd import capi
    rng = np.random.default_rng(H * W + Cin)
    B = 2
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    b = rng.standard_normal((Cout,)).astype(np.float32)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    dx, dout = _dev(ctx, x), ctx.alloc(B * Ho * Wo * Cout * 4)
    ctx._chk(capi.lib.rfe_k_conv3x3(ctx.h, dx.ptr, B, H, W, Cin, w.ctypes.data, b.ctypes.data, Cout, relu, pool, dout.ptr))
    got = dout.download((B, Ho, Wo, Cout), np.float32)
    for i in range(B):
        ref = oracle.conv3x3(x[i], w, b, relu=bool(relu), pool=bool(pool))
        assert np.array_equal(got[i], ref), f"frame {i}: max diff {np.abs(got[i] - ref).max()}"
    dx.free(); dout.free()


# ------------------------------------------------------------------ SuperPoint
@pytest.mark.parametrize("H,W", [(64, 96), (120, 160), (72, 200), (77, 101), (63, 130), (100, 150)])
def test_superpoint_maps_bitexact(ctx, oracle, H, W):
    """includes sizes that are not multiples of 8 (and of 2, 4): floor pooling, maps on the 8*(H/8) x 8*(W/8) frame"""
    from rover_slam_amd import capi
    frames, _ = synth.make_frames(2, H, W, seed=H + W)
    dimg = _dev(ctx, frames)
    Hs, Ws = H // 8 * 8, W // 8 * 8
    ds, dn, dd = ctx.alloc(2 * Hs * Ws * 4), ctx.alloc(2 * Hs * Ws * 4), ctx.alloc(2 * (H // 8) * (W // 8) * 256 * 4)
    ctx._chk(capi.lib.rfe_k_scoremap(ctx.h, dimg.ptr, H, W, W, 2, ds.ptr, dn.ptr, dd.ptr))
    smap = ds.download((2, Hs, Ws), np.float32)
    nmap = dn.download((2, Hs, Ws), np.float32)
    dmap = dd.download((2, H // 8, W // 8, 256), np.float32)
    w = Wt.make_superpoint(seed=7)
    for i in range(2):
        r = oracle.superpoint(w, frames[i], kmax=16, debug=True)
        assert np.array_equal(smap[i], r["scoremap"])
        assert np.array_equal(nmap[i], r["nms"])
        assert np.array_equal(dmap[i], r["descmap"])
    for d in (dimg, ds, dn, dd):
        d.free()


@pytest.mark.parametrize("H,W,kmax", [(120, 160, 4096), (120, 160, 100), (64, 96, 33), (240, 320, 512), (125, 163, 4096), (93, 201, 64),
                                      (480, 640, 700), (256, 400, 300), (488, 304, 128)])   # 60x80 / 32x50 / 61x38 grids: composite conv tiles
def test_extract_bitexact_vs_oracle(ctx, oracle, H, W, kmax):
    frames, _ = synth.make_frames(3, H, W, seed=kmax)
    n, kxy, score, desc = ctx.extract(frames, kmax=kmax)
    w = Wt.make_superpoint(seed=7)
    for i in range(3):
        r = oracle.superpoint(w, frames[i], kmax=kmax)
        assert n[i] == r["n"]
        assert np.array_equal(kxy[i], r["kxy"])        # keypoint indices bit-exact, same order
        assert np.array_equal(score[i], r["score"])    # scores bit-exact
        assert np.array_equal(desc[i], r["desc"])      # stated bar 1e-4; achieved: bit-exact


@pytest.mark.parametrize("B,H,W,kmax", [(1, 1080, 1920, 4096), (2, 720, 1280, 2048)])
def test_extract_large_frames_bitexact_vs_oracle(ctx, oracle, B, H, W, kmax):
    """frames well above the benchmark's 640 x 480 (the reference accepts any size its ONNX graph does, superpoint_onnx.cc:100: dynamic H, W): one
    1920 x 1080 frame (135 x 240 cells, 530 MB of conv1 activations) and two 1280 x 720 frames, the largest keypoint budgets -- keypoints, scores and
    descriptors bit for bit"""
    frames, _ = synth.make_frames(B, H, W, seed=H + kmax)
    n, kxy, score, desc = ctx.extract(frames, kmax=kmax)
    w = Wt.make_superpoint(seed=7)
    for i in range(B):
        r = oracle.superpoint(w, frames[i], kmax=kmax)
        assert n[i] == r["n"] and r["n"] > 1000
        assert np.array_equal(kxy[i], r["kxy"]) and np.array_equal(score[i], r["score"]) and np.array_equal(desc[i], r["desc"])


def test_extract_degenerate_frames_bitexact_vs_oracle(ctx, oracle):
    """what a camera really delivers now and then: an all-black frame, a saturated one, a frame whose lower half is constant.  A constant region makes the
    score map constant per cell position, so simple_nms's EQUALITY mask keeps every pixel of it and top-k has to break thousands of exact ties (score descending,
    pixel index ascending: the published graph's TopK on the row-major candidate list) -- same keypoints, same order, same descriptors as the oracle"""
    H, W, kmax = 240, 320, 1024
    frames, _ = synth.make_frames(4, H, W, seed=5)
    frames[0] = 0
    frames[1] = 255
    frames[2, H // 2:] = 37
    n, kxy, score, desc = ctx.extract(frames, kmax=kmax)
    w = Wt.make_superpoint(seed=7)
    for i in range(4):
        r = oracle.superpoint(w, frames[i], kmax=kmax)
        assert n[i] == r["n"], (i, n[i], r["n"])
        assert np.array_equal(kxy[i], r["kxy"]) and np.array_equal(score[i], r["score"]) and np.array_equal(desc[i], r["desc"]), i


@pytest.mark.parametrize("H,W,kmax", [(120, 160, 300), (480, 640, 1024), (93, 201, 64)])
def test_extract_float_entry_bitexact_vs_oracle(ctx, oracle, H, W, kmax):
    """rfe_extract_f32 = the reference's Extractor_Inference on a CV_32F image (superpoint_onnx.cc:88-118): (a) on u8 * (1/255) it is the u8
    entry bit for bit, (b) on values off the 1/255 lattice and outside [0, 1] it is the oracle's float entry bit for bit."""
    frames, _ = synth.make_frames(2, H, W, seed=kmax + 1)
    w = Wt.make_superpoint(seed=7)
    f = frames.astype(np.float32) * np.float32(0.003921568859368563)
    a = ctx.extract(frames, kmax=kmax)
    b = ctx.extract_f32(f, kmax=kmax)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    rng = np.random.default_rng(kmax)
    g = (f * np.float32(1.7) - np.float32(0.2) + rng.uniform(-0.001, 0.001, f.shape).astype(np.float32)).astype(np.float32)
    n, kxy, score, desc = ctx.extract_f32(g, kmax=kmax)
    for i in range(2):
        r = oracle.superpoint(w, g[i], kmax=kmax)
        assert n[i] == r["n"] and r["n"] > 10
        assert np.array_equal(kxy[i], r["kxy"]) and np.array_equal(score[i], r["score"]) and np.array_equal(desc[i], r["desc"])


@pytest.mark.parametrize("B,H,W", [(2, 480, 640), (2, 480, 752), (4, 240, 320), (2, 376, 1241)])
def test_extract_few_frames_one_round_nms_tiles(ctx, oracle, B, H, W):
    """Two to four frames per call (a pair, a stereo frame): the fused NMS picks its tile height (32 / 40 / 48 rows) so that all frames fit one
    round of workgroups, the selection is the rank-all form.  Every frame bit-exact against the oracle."""
    frames, _ = synth.make_frames(B, H, W, seed=B * H + W)
    n, kxy, score, desc = ctx.extract(frames, kmax=1024)
    w = Wt.make_superpoint(seed=7)
    for i in range(B):
        r = oracle.superpoint(w, frames[i], kmax=1024)
        assert n[i] == r["n"]
        assert np.array_equal(kxy[i], r["kxy"]) and np.array_equal(score[i], r["score"]) and np.array_equal(desc[i], r["desc"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_extract_vs_golden(ctx, golden_dir, tag):
    g = np.load(f"{golden_dir}/sp_{tag}.npz")
    n, kxy, score, desc = ctx.extract(g["image"], kmax=4096)
    k = int(g["n"])
    assert n[0] == k
    assert np.array_equal(kxy[0, :k], g["kxy"])
    assert np.abs(score[0, :k] - g["score"]).max() < 2e-5
    assert np.abs(desc[0, :k] - g["desc"]).max() < 1e-4   # north_star tolerance


def test_extract_variable_k_and_dustbin(oracle):
    """second weight set (dustbin bias +6): K < Kmax, differs per frame; padding rows are zero."""
    from rover_slam_amd import capi
    c = capi.Context(0)
    w = Wt.make_superpoint(seed=9, dustbin_bias=6.0)
    c.set_weights(capi.KIND_SUPERPOINT, w)
    frames, _ = synth.make_frames(4, 96, 128, seed=77)
    n, kxy, score, desc = c.extract(frames, kmax=512)
    for i in range(4):
        r = oracle.superpoint(w, frames[i], kmax=512)
        assert n[i] == r["n"] and 0 < n[i] < 512
        assert np.array_equal(kxy[i], r["kxy"]) and np.array_equal(score[i], r["score"])
        assert np.array_equal(desc[i], r["desc"])
        assert not desc[i, n[i]:].any()
    c.close()


def test_extract_batch_equals_single(ctx):
    frames, _ = synth.make_frames(5, 120, 160, seed=5)
    nb, kb, sb, db = ctx.extract(frames, kmax=256)
    for i in range(5):
        n1, k1, s1, d1 = ctx.extract(frames[i], kmax=256)
        assert n1[0] == nb[i] and np.array_equal(k1[0], kb[i]) and np.array_equal(s1[0], sb[i]) and np.array_equal(d1[0], db[i])


def test_full_size_batch_equals_single(ctx):
    """640x480, batch of 12: the throughput tilings (8x32 conv tile, wide 4x80 and 12x16 tiles on the 60x80 layers,
    128x256 GEMM tiles) against the latency tilings a single frame takes (4x32x32 conv tile, 64-row GEMM tiles) --
    different tile shapes, same reduction order, so every output must be bit-identical (and the single-frame path is
    the one test_full_size_640x480_vs_oracle pins to the oracle)."""
    frames, _ = synth.make_frames(12, 480, 640, seed=77)
    nb, kb, sb, db = ctx.extract(frames, kmax=1024)
    for i in (0, 5, 11):
        n1, k1, s1, d1 = ctx.extract(frames[i], kmax=1024)
        assert n1[0] == nb[i] and np.array_equal(k1[0], kb[i]) and np.array_equal(s1[0], sb[i]) and np.array_equal(d1[0], db[i])


# ------------------------------------------------------------------ LightGlue
def _pair_from_golden(g):
    return g["k0n"], g["k1n"], g["d0"], g["d1"]


@pytest.mark.parametrize("tag", ["a", "b"])
def test_lightglue_vs_oracle_and_golden(ctx, oracle, golden_dir, tag):
    from rover_slam_amd import capi
    g = np.load(f"{golden_dir}/lg_{tag}.npz")
    k0, k1, d0, d1 = _pair_from_golden(g)
    M, N = k0.shape[0], k1.shape[0]
    bufs = [_dev(ctx, a) for a in (k0, k1, d0, d1)]
    dx0, dx1, dsc = ctx.alloc(M * 1024), ctx.alloc(N * 1024), ctx.alloc(M * N * 4)
    ctx._chk(capi.lib.rfe_k_lightglue_taps(ctx.h, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, M, N, dx0.ptr, dx1.ptr, dsc.ptr))
    x0, x1 = dx0.download((M, 256), np.float32), dx1.download((N, 256), np.float32)
    sc = dsc.download((M, N), np.float32)
    r = oracle.lightglue(Wt.make_lightglue(seed=11), k0, k1, d0, d1, debug=True)
    # fp32 tolerance: flash-style online softmax and permuted PV reduction order vs the oracle's plain softmax
    assert np.abs(x0 - r["x0"]).max() < LG_STATE_TOL and np.abs(x1 - r["x1"]).max() < LG_STATE_TOL
    assert np.abs(sc - r["scores"]).max() < LG_LOGSCORE_RTOL * np.abs(r["scores"]).max()   # log-domain assignment scores
    S, pairs, ms = ctx.match(k0[None], k1[None], d0[None], d1[None], [M], [N])
    assert S[0] == r["S"] == len(g["pairs"])
    assert np.array_equal(pairs[0, :S[0]], r["pairs"]) and np.array_equal(pairs[0, :S[0]], g["pairs"])
    assert np.abs(ms[0, :S[0]] - r["ms"]).max() < LG_SCORE_TOL_SMALL
    assert np.abs(ms[0, :S[0]] - g["ms"]).max() < LG_SCORE_TOL_SMALL
    for b in bufs + [dx0, dx1, dsc]:
        b.free()


@pytest.mark.parametrize("tag", ["c", "d"])
@pytest.mark.parametrize("fold", [1, 0])
def test_lightglue_fullsize_vs_oracle_and_golden(ctx, oracle, golden_dir, tag, fold):
    """K = 1024 (c) and the ragged 700 x 1024 pair (d), several hundred matches each: HIP path (with and without the
    Wo fold) against the oracle and against the independent HF fixture -- final token states, match lists, scores."""
    from rover_slam_amd import capi
    g = np.load(f"{golden_dir}/lg_{tag}.npz")
    k0, k1, d0, d1 = _pair_from_golden(g)
    M, N = k0.shape[0], k1.shape[0]
    ctx.set_option(capi.OPT_LG_FOLD_WO, fold)
    try:
        bufs = [_dev(ctx, a) for a in (k0, k1, d0, d1)]
        dx0, dx1 = ctx.alloc(M * 1024), ctx.alloc(N * 1024)
        ctx._chk(capi.lib.rfe_k_lightglue_taps(ctx.h, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, M, N, dx0.ptr, dx1.ptr, None))
        x0, x1 = dx0.download((M, 256), np.float32), dx1.download((N, 256), np.float32)
        S, pairs, ms = ctx.match(k0[None], k1[None], d0[None], d1[None], [M], [N])
    finally:
        ctx.set_option(capi.OPT_LG_FOLD_WO, 1)
    r = oracle.lightglue(Wt.make_lightglue(seed=11), k0, k1, d0, d1, debug=True)
    assert np.abs(x0 - r["x0"]).max() < LG_STATE_TOL and np.abs(x1 - r["x1"]).max() < LG_STATE_TOL
    assert np.abs(x0[::4] - g["x0_rows4"]).max() < LG_STATE_TOL and np.abs(x1[::4] - g["x1_rows4"]).max() < LG_STATE_TOL
    for ref_pairs, ref_ms in ((r["pairs"], r["ms"]), (g["pairs"], g["ms"])):
        ok, dev = lists_agree(pairs[0, :S[0]], ms[0, :S[0]], ref_pairs, ref_ms)
        assert ok and dev < LG_SCORE_TOL, (ok, dev)
    assert S[0] > 400
    for b in bufs + [dx0, dx1]:
        b.free()


def test_extract_fullsize_topk_vs_golden(ctx, golden_dir):
    """480x640 (bench frame 0) and 480x752 through the top-k path against the independent HF fixture: the same 1024 of
    the > 5000 candidates, scores and descriptors keypoint by keypoint (the oracle is bit-exact with the HIP path, the
    fixture is a different fp32 code base: 5e-6 / 2e-6)."""
    for tag in ("d", "e"):
        g = np.load(f"{golden_dir}/sp_{tag}.npz")
        n, kxy, score, desc = ctx.extract(g["image"], kmax=1024)
        where = {tuple(k): i for i, k in enumerate(g["kxy"])}
        assert n[0] == 1024 and {tuple(k) for k in kxy[0]} == set(where)
        perm = [where[tuple(k)] for k in kxy[0]]
        assert np.abs(score[0] - g["score"][perm]).max() < 5e-6 and np.abs(desc[0] - g["desc"][perm]).max() < 2e-6


@pytest.mark.parametrize("H,W", [(8, 8), (9, 17), (16, 8), (24, 40), (15, 15), (8, 200), (200, 8), (33, 31), (94, 310)])
def test_extract_tiny_and_thin_images(ctx, oracle, H, W):
    """The smallest inputs the entry points accept (one feature cell) and strips a single tile wide or tall, batch 1 and 3."""
    w = Wt.make_superpoint(seed=7)
    for B in (1, 3):
        fr, _ = synth.make_frames(B, max(H, 48), max(W, 48), seed=H * 100 + W)
        fr = np.ascontiguousarray(fr[:, :H, :W])
        n, kxy, score, desc = ctx.extract(fr, kmax=64)
        for i in range(B):
            r = oracle.superpoint(w, fr[i], kmax=64)
            k = r["n"]
            assert n[i] == k
            assert np.array_equal(kxy[i, :k], r["kxy"][:k]) and np.array_equal(score[i, :k], r["score"][:k]) and np.array_equal(desc[i, :k], r["desc"][:k])


def test_extract_batch13_composite_conv_tiles(ctx, oracle):
    """13 frames of 480 x 640 in one call: all four 60 x 80 layers take the composite 8 x 32 tiles over the two-frames-wide canvas
    (7 frame pairs, the last one half empty; tiles assembled from pieces of up to four frames).  Every frame bit-exact."""
    frames, _ = synth.make_frames(13, 480, 640, seed=1313)
    n, kxy, score, desc = ctx.extract(frames, kmax=256)
    w = Wt.make_superpoint(seed=7)
    for i in range(13):
        r = oracle.superpoint(w, frames[i], kmax=256)
        assert n[i] == r["n"] == 256
        assert np.array_equal(kxy[i], r["kxy"]) and np.array_equal(score[i], r["score"]) and np.array_equal(desc[i], r["desc"]), f"frame {i}"


def test_extract_kitti_size_vs_oracle_and_golden(ctx, oracle, golden_dir):
    """KITTI's 1241 x 376 (not a multiple of 8; the reference graph has dynamic axes): bit-exact against the oracle, and against the
    independent HF fixture sp_f (top-k of 7790 candidates) / sp_g (101 x 151, all candidates)."""
    w = Wt.make_superpoint(seed=7)
    for tag in ("f", "g"):
        g = np.load(f"{golden_dir}/sp_{tag}.npz")
        kmax = int(g["kmax"])
        n, kxy, score, desc = ctx.extract(g["image"], kmax=kmax)
        r = oracle.superpoint(w, g["image"], kmax=kmax)
        assert n[0] == r["n"] == int(g["n"])
        k = r["n"]
        assert np.array_equal(kxy[0, :k], r["kxy"][:k]) and np.array_equal(score[0, :k], r["score"][:k]) and np.array_equal(desc[0, :k], r["desc"][:k])
        where = {tuple(q): i for i, q in enumerate(g["kxy"])}
        assert {tuple(q) for q in kxy[0, :k]} == set(where)
        perm = [where[tuple(q)] for q in kxy[0, :k]]
        assert np.abs(score[0, :k] - g["score"][perm]).max() < 5e-6 and np.abs(desc[0, :k] - g["desc"][perm]).max() < 2e-6


def test_lightglue_ragged_batch(ctx, oracle):
    """P=3 pairs with different (m,n) padded to a common Mmax/Nmax; includes an empty side."""
    rng = np.random.default_rng(21)
    Mmax, Nmax = 150, 131
    sizes = [(150, 131), (70, 131), (97, 40)]
    P = len(sizes)
    k0 = np.zeros((P, Mmax, 2), np.float32); k1 = np.zeros((P, Nmax, 2), np.float32)
    d0 = np.zeros((P, Mmax, 256), np.float32); d1 = np.zeros((P, Nmax, 256), np.float32)
    for p, (m, n) in enumerate(sizes):
        a = rng.standard_normal((max(m, n), 256)).astype(np.float32)
        a /= np.linalg.norm(a, axis=1, keepdims=True)
        kk = rng.uniform(-0.9, 0.9, (max(m, n), 2)).astype(np.float32)
        perm = rng.permutation(max(m, n))
        d0[p, :m] = a[:m]; k0[p, :m] = kk[:m]
        bb = a[perm][:n] + 0.05 * rng.standard_normal((n, 256)).astype(np.float32)
        d1[p, :n] = bb / np.linalg.norm(bb, axis=1, keepdims=True); k1[p, :n] = kk[perm][:n]
    S, pairs, ms = ctx.match(k0, k1, d0, d1, [s[0] for s in sizes], [s[1] for s in sizes])
    w = Wt.make_lightglue(seed=11)
    for p, (m, n) in enumerate(sizes):
        r = oracle.lightglue(w, k0[p, :m], k1[p, :n], d0[p, :m], d1[p, :n])
        assert S[p] == r["S"]
        assert np.array_equal(pairs[p, :S[p]], r["pairs"])
        assert np.abs(ms[p, :S[p]] - r["ms"]).max() < LG_SCORE_TOL_SMALL if S[p] else True


def test_lightglue_k1024_batch_equals_single(ctx):
    """K = 1024: six pairs in one call (128x256 GEMM tiles, one attention workgroup per query block) against the same
    pairs one at a time (64-row GEMM tiles, split-key attention + combine).  Same GEMM reduction order; the attention
    merges its key ranges in a different order, so scores agree to the stated 5e-4 and the match lists exactly."""
    rng = np.random.default_rng(33)
    P, K = 6, 1024
    lens0 = [1024, 1024, 700, 1024, 333, 1024]; lens1 = [1024, 900, 1024, 257, 1024, 1024]
    k0 = np.zeros((P, K, 2), np.float32); k1 = np.zeros((P, K, 2), np.float32)
    d0 = np.zeros((P, K, 256), np.float32); d1 = np.zeros((P, K, 256), np.float32)
    for p in range(P):
        a = rng.standard_normal((K, 256)).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
        kk = rng.uniform(-0.9, 0.9, (K, 2)).astype(np.float32)
        perm = rng.permutation(K)
        m, n = lens0[p], lens1[p]
        d0[p, :m] = a[:m]; k0[p, :m] = kk[:m]
        bb = a[perm][:n] + 0.05 * rng.standard_normal((n, 256)).astype(np.float32)
        d1[p, :n] = bb / np.linalg.norm(bb, axis=1, keepdims=True); k1[p, :n] = (kk[perm][:n] + 0.002).astype(np.float32)
    S, pairs, ms = ctx.match(k0, k1, d0, d1, lens0, lens1)
    assert (S > 50).all()
    for p in range(P):
        S1, p1, m1 = ctx.match(k0[p:p + 1], k1[p:p + 1], d0[p:p + 1], d1[p:p + 1], lens0[p:p + 1], lens1[p:p + 1])
        assert S1[0] == S[p] and np.array_equal(p1[0, :S1[0]], pairs[p, :S[p]])
        assert np.abs(m1[0, :S1[0]] - ms[p, :S[p]]).max() < 5e-4


def test_match_fused_semantics(ctx, oracle):
    """rfe_match_fused == NormalizeKeypoints + LightGlue + Matcher_PostProcess_fused, incl. the 300x400 quirk."""
    frames, _ = synth.make_frames(2, 120, 160, seed=3)
    n, kxy, score, desc = ctx.extract(frames, kmax=128)
    kp0, kp1 = kxy[0, :n[0]].astype(np.float32), kxy[1, :n[1]].astype(np.float32)
    w = Wt.make_lightglue(seed=11)
    for rows, cols in ((120, 160), (300, 400)):
        size, vn = ctx.match_fused(kp0, kp1, desc[0, :n[0]], desc[1, :n[1]], rows, cols)
        r = oracle.lightglue(w, oracle.normalize_keypoints(kp0, rows, cols), oracle.normalize_keypoints(kp1, rows, cols),
                             desc[0, :n[0]], desc[1, :n[1]])
        size_ref, vn_ref = oracle.postprocess_fused(r["pairs"], r["ms"], 0.0, int(n[0]))
        assert size == size_ref and np.array_equal(vn, vn_ref)


# ------------------------------------------------------------------ full size properties
def test_full_size_properties(ctx):
    """640x480 (BASELINE size): size-independent properties instead of the (slow) oracle."""
    frames, offs = synth.make_frames(3, 480, 640, seed=20240314)
    n, kxy, score, desc = ctx.extract(frames, kmax=1024)
    for i in range(3):
        k = n[i]
        assert 0 < k <= 1024
        xy = kxy[i, :k]
        assert (xy[:, 0] >= 4).all() and (xy[:, 0] < 636).all() and (xy[:, 1] >= 4).all() and (xy[:, 1] < 476).all()
        assert (score[i, :k] > 0.0005).all()
        if k == 1024:
            assert (np.diff(score[i, :k]) <= 0).all()           # sortedness of the top-k path
        assert np.allclose(np.linalg.norm(desc[i, :k], axis=1), 1.0, atol=1e-5)
        flat = xy[:, 1] * 640 + xy[:, 0]
        assert len(np.unique(flat)) == k
    # idempotence
    n2, kxy2, score2, desc2 = ctx.extract(frames, kmax=1024)
    assert np.array_equal(kxy, kxy2) and np.array_equal(desc, desc2)
    # matching consecutive frames: mutual, unique, and geometrically consistent with the known shift
    kp = [kxy[i, :n[i]].astype(np.float32) for i in range(3)]
    size, vn = ctx.match_fused(kp[0], kp[1], desc[0, :n[0]], desc[1, :n[1]], 480, 640)
    j = vn[vn >= 0]
    assert size == len(j) and len(np.unique(j)) == len(j)


# ------------------------------------------------------------------ shapes / concurrency / errors
def test_extract_752x480_stereo_size(ctx, oracle):
    """EuRoC native size (BASELINE config 5): W = 752 is not a multiple of the 32-px conv tile."""
    frames, _ = synth.make_frames(2, 480, 752, seed=752)
    n, kxy, score, desc = ctx.extract(frames, kmax=1024)
    w = Wt.make_superpoint(seed=7)
    r = oracle.superpoint(w, frames[1], kmax=1024)
    assert n[1] == r["n"] and np.array_equal(kxy[1], r["kxy"])
    assert np.array_equal(score[1], r["score"]) and np.array_equal(desc[1], r["desc"])


def test_two_contexts_concurrently(oracle):
    """Left / right extractors of a stereo frame run in two threads with one session each in the reference
    (src/Frame.cc:142-147): two ctxs on one device must work concurrently and agree with the oracle."""
    import threading
    from rover_slam_amd import capi
    w = Wt.make_superpoint(seed=7)
    frames, _ = synth.make_frames(2, 120, 160, seed=99)
    out = [None, None]

    def work(i):
        c = capi.Context(0)
        c.set_weights(capi.KIND_SUPERPOINT, w)
        for _ in range(5):
            out[i] = c.extract(frames[i], kmax=256)
        c.close()

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(2):
        r = oracle.superpoint(w, frames[i], kmax=256)
        n, kxy, score, desc = out[i]
        assert n[0] == r["n"] and np.array_equal(kxy[0], r["kxy"]) and np.array_equal(desc[0], r["desc"])


@pytest.mark.parametrize("B,H,W,K", [(5, 120, 160, 128), (3, 200, 152, 48), (4, 120, 160, 300), (2, 480, 640, 1024)])
def test_extract_is_repeatable_when_the_heads_overlap(ctx, oracle, B, H, W, K):
    """Regression (round 5): the descriptor head runs on a side stream next to the detector head; with the 16-channel form of conv3x3_t16d_kernel the heads
    overlapped differently and a latent race showed -- an LDS read hoisted by the machine scheduler above the barrier that orders it behind its LDS-DMA copy
    (tests/test_isa_screen.py is the static screen).  Ten calls must give identical bytes, and the oracle's."""
    frames, _ = synth.make_frames(B, H, W, seed=11)
    first = ctx.extract(frames, kmax=K)
    for _ in range(9):
        again = ctx.extract(frames, kmax=K)
        assert all(np.array_equal(a, b) for a, b in zip(first, again))
    w = Wt.make_superpoint(seed=7)
    for i in (0, B - 1):
        r = oracle.superpoint(w, frames[i], kmax=K)
        assert first[0][i] == r["n"] and np.array_equal(first[1][i], r["kxy"]) and np.array_equal(first[3][i], r["desc"])


@pytest.mark.parametrize("H,W,K,bias", [(120, 160, 128, None), (136, 200, 1024, None), (480, 640, 1024, None), (240, 320, 1024, 9.5), (101, 77, 64, None)])
def test_fused_detector_tail_equals_the_separate_launches(ctx, H, W, K, bias):
    """Round 6: calls of up to four frames run the detector tail as sp_tail_lat_kernel + select_rankall_keys_kernel (softmax from the logits, NMS and an UNORDERED
    candidate list in one launch; rank-determined outputs), larger calls the separate softmax / NMS / count / compact / select launches.  The same frames through both
    -- five in one call, then the first four, two and one of them -- must give identical bytes per frame: counts, keypoints in order, scores, descriptors.  Sizes that
    pick each tile height (32 / 40 / 48 rows), a size that is no multiple of 8, a budget above the candidate count of small frames, and the dustbin-biased weights whose
    candidate count stays BELOW Kmax (row-major order, ranked by pixel index in the fused form)."""
    from rover_slam_amd import capi
    if bias is not None:
        ctx.set_weights(capi.KIND_SUPERPOINT, Wt.make_superpoint(seed=7, dustbin_bias=bias))
    try:
        frames, _ = synth.make_frames(5, H, W, seed=H + W)
        ref = ctx.extract(frames, kmax=K)                                  # 5 frames: the separate launches
        for nb in (4, 2, 1):
            got = ctx.extract(frames[:nb], kmax=K)                         # <= 4 frames: the fused tail
            for a, b in zip(ref, got):
                assert np.array_equal(a[:nb], b), (nb, a.shape)
        if bias is not None:
            assert 0 < int(ref[0].min()) and int(ref[0].max()) < K        # the case really exercises K < Kmax
    finally:
        if bias is not None:
            ctx.set_weights(capi.KIND_SUPERPOINT, Wt.make_superpoint(seed=7))


def test_tiny_and_odd_frames_vs_oracle(ctx, oracle):
    """The reference graph has dynamic axes: any frame of at least 8 x 8 pixels goes.  Frames from ONE cell (8 x 8) up to just past a tile of the fused
    detector tail (72 x 104 haloed), odd sizes, random noise (tie-rich), batches on both sides of the four-frame switch between the fused tail and the separate
    launches, budgets of 1, 7 and 4096 keypoints: every output bit for bit."""
    w = Wt.make_superpoint(seed=7)
    rng = np.random.default_rng(0)
    for (H, W) in [(8, 8), (9, 15), (16, 8), (8, 64), (24, 40), (31, 33), (47, 9), (64, 64), (65, 130), (73, 105)]:
        for B in (1, 4, 5):
            frames = rng.integers(0, 256, (B, H, W), dtype=np.uint8)
            for K in (1, 7, 4096):
                n, kxy, score, desc = ctx.extract(frames, kmax=K)
                for i in (0, B - 1):
                    r = oracle.superpoint(w, frames[i], kmax=K)
                    assert n[i] == r["n"] and np.array_equal(kxy[i], r["kxy"]) and np.array_equal(score[i], r["score"]) and np.array_equal(desc[i], r["desc"]), (H, W, B, K, i)


@pytest.mark.parametrize("border,always,thr,bias", [(2, 1, 0.0005, None), (0, 0, 0.005, None), (7, 1, 0.0005, 9.5), (4, 1, 0.0005, 9.5)])
def test_fused_detector_tail_graph_hyper_parameters_vs_oracle(ctx, oracle, border, always, thr, bias):
    """The graph's constants on the fused tail's side of the switch (published NMS radius 4, <= 4 frames): border width, the unconditional top-k (counts below
    Kmax then come out in score order instead of row-major -- the ranking kernel's two orders), another detection threshold; against the oracle with the same
    constants, and the same frames in a five-frame call (separate launches)."""
    from rover_slam_amd import capi
    w = Wt.make_superpoint(seed=7, dustbin_bias=bias) if bias is not None else Wt.make_superpoint(seed=7)
    ctx.set_weights(capi.KIND_SUPERPOINT, w)
    ctx.set_hparams(sp_nms_radius=4, sp_remove_borders=border, sp_topk_always=always)
    try:
        frames, _ = synth.make_frames(5, 136, 200, seed=border * 10 + always)
        K = 512
        got = ctx.extract(frames[:3], kmax=K, thr=thr)
        five = ctx.extract(frames, kmax=K, thr=thr)
        for a, b in zip(got, five):
            assert np.array_equal(a, b[:3])
        for i in range(3):
            r = oracle.superpoint(w, frames[i], kmax=K, thr=thr, nms_radius=4, border=border, topk_always=bool(always))
            assert got[0][i] == r["n"] and np.array_equal(got[1][i], r["kxy"]) and np.array_equal(got[2][i], r["score"]) and np.array_equal(got[3][i], r["desc"]), i
        if bias is not None:
            assert int(got[0].max()) < K          # the unconditional top-k really had fewer candidates than Kmax to order
    finally:
        ctx.set_hparams(sp_nms_radius=4, sp_remove_borders=4, sp_topk_always=0)
        ctx.set_weights(capi.KIND_SUPERPOINT, Wt.make_superpoint(seed=7))


def test_host_graph_option_gives_identical_results(ctx, oracle):
    """RFE_OPT_HOST_GRAPH: the host entries replay a captured hipGraph per call shape (captured on the third call of a shape, four shapes kept; weights, hyper-parameters, options, shape
    and workspace addresses are part of the key).  Same bytes as ordinary launches for extract (u8 / float / binarised) and match, across shape changes,
    a weight change, a hyper-parameter change and back."""
    from rover_slam_amd import capi
    fa, _ = synth.make_frames(2, 120, 160, seed=41)
    fb, _ = synth.make_frames(1, 240, 320, seed=42)
    w2 = Wt.make_superpoint(seed=9, dustbin_bias=7.0)

    def run_all():
        out = []
        for fr, K in ((fa, 200), (fb, 512), (fa, 200), (fa, 64)):
            out.append(ctx.extract(fr, kmax=K, binarized=True))
            out.append(ctx.extract_f32(fr.astype(np.float32) * np.float32(1.0 / 255.0), kmax=K))
        n, kxy, _, desc, _ = out[0]
        k0 = oracle.normalize_keypoints(kxy[0, :n[0]].astype(np.float32), 120, 160); k1 = oracle.normalize_keypoints(kxy[1, :n[1]].astype(np.float32), 120, 160)
        for _ in range(3):
            out.append(ctx.match(k0[None], k1[None], desc[0, :n[0]][None], desc[1, :n[1]][None], [int(n[0])], [int(n[1])]))
        ctx.set_weights(capi.KIND_SUPERPOINT, w2)
        for _ in range(3):
            out.append(ctx.extract(fa, kmax=200))
        ctx.set_weights(capi.KIND_SUPERPOINT, Wt.make_superpoint(seed=7))
        ctx.set_hparams(sp_nms_radius=3, sp_remove_borders=2, sp_topk_always=1)
        for _ in range(3):
            out.append(ctx.extract(fa, kmax=200))
        ctx.set_hparams(sp_nms_radius=4, sp_remove_borders=4, sp_topk_always=0)
        for _ in range(3):
            out.append(ctx.extract(fa, kmax=200))
        return out

    plain = run_all()
    ctx.set_option(capi.OPT_HOST_GRAPH, 1)
    try:
        assert ctx.get_option(capi.OPT_HOST_GRAPH) == 1
        graphed = run_all()
        graphed2 = run_all()
    finally:
        ctx.set_option(capi.OPT_HOST_GRAPH, 0)
    for a, b, c2 in zip(plain, graphed, graphed2):
        S_or_n = a[0]
        for x, y, z in zip(a, b, c2):
            if x.ndim >= 2 and x.shape[:1] == S_or_n.shape and len(a) == 3:      # match outputs: rows beyond S are undefined
                for q in range(len(S_or_n)):
                    assert np.array_equal(x[q, :S_or_n[q]], y[q, :S_or_n[q]]) and np.array_equal(x[q, :S_or_n[q]], z[q, :S_or_n[q]])
            else:
                assert np.array_equal(x, y) and np.array_equal(x, z)
    assert not np.array_equal(plain[-4][1], plain[-7][1])            # the hyper-parameter change really changed the output (and the graph key)


def test_extract_into_pinned_host_memory_is_identical(ctx):
    """rfe_host_malloc (round 5): a descriptor output inside a block from it is written by the DMA engine directly instead of being staged through the ctx's
    pinned block and copied on the host -- same bytes, for the u8, the binarised and the float entry, with one and with three frames; a pointer INTO the block
    (offset) counts, a block that is too small does not (falls back to staging)."""
    import ctypes as C
    from rover_slam_amd import capi
    frames, _ = synth.make_frames(3, 120, 160, seed=31)
    K = 300
    for B in (1, 3):
        n0, k0, s0, d0, b0 = ctx.extract(frames[:B], kmax=K, binarized=True)
        nf, kf, sf, df = ctx.extract_f32(frames[:B].astype(np.float32) * np.float32(1.0 / 255.0), kmax=K)
        nbytes = B * K * 1024
        blk = C.c_void_p()
        assert capi.lib.rfe_host_malloc(nbytes + 4096, C.byref(blk)) == 0 and blk.value
        try:
            for off in (0, 1024):
                desc = np.ctypeslib.as_array(C.cast(blk.value + off, C.POINTER(C.c_float)), shape=(B, K, 256))
                n = np.zeros((B,), np.int32); kxy = np.zeros((B, K, 2), np.int32); sc = np.zeros((B, K), np.float32); dbin = np.zeros((B, K, 256), np.uint8)
                desc[:] = -7.0
                ctx._chk(capi.lib.rfe_extract_u8_bin(ctx.h, frames[:B].ctypes.data, 120, 160, 160, B, K, 0.0005, n.ctypes.data, kxy.ctypes.data, sc.ctypes.data,
                                                     blk.value + off, dbin.ctypes.data))
                assert np.array_equal(n, n0) and np.array_equal(kxy, k0) and np.array_equal(sc, s0) and np.array_equal(desc, d0) and np.array_equal(dbin, b0)
                desc[:] = -7.0
                img = np.ascontiguousarray(frames[:B].astype(np.float32) * np.float32(1.0 / 255.0))
                ctx._chk(capi.lib.rfe_extract_f32(ctx.h, img.ctypes.data, 120, 160, 160, B, K, 0.0005, n.ctypes.data, kxy.ctypes.data, sc.ctypes.data, blk.value + off))
                assert np.array_equal(n, nf) and np.array_equal(desc, df)
        finally:
            capi.lib.rfe_host_free(blk)
    assert capi.lib.rfe_host_malloc(0, C.byref(blk)) != 0            # bad argument, not a crash
    capi.lib.rfe_host_free(None)                                      # no-op


def test_error_paths(ctx):
    from rover_slam_amd import capi
    c = capi.Context(0)
    img = np.zeros((1, 64, 96), np.uint8)
    with pytest.raises(capi.RfeError, match="weights not loaded"):
        c.extract(img)
    c.set_weights(capi.KIND_SUPERPOINT, Wt.make_superpoint(seed=7))
    with pytest.raises(capi.RfeError, match="at least 8"):
        c.extract(np.zeros((1, 7, 100), np.uint8))
    with pytest.raises(capi.RfeError, match="wrong float count"):
        c.set_weights(capi.KIND_LIGHTGLUE, np.zeros(10, np.float32))
    with pytest.raises(capi.RfeError):
        c.load_weights(sp_path="/nonexistent/sp.rfew")
    # empty sides: no crash, no matches (reference would index an empty ORT output, lightglue_onnx.cpp:404)
    size, vn = ctx.match_fused(np.zeros((0, 2), np.float32), np.zeros((5, 2), np.float32), np.zeros((0, 256), np.float32),
                               np.zeros((5, 256), np.float32), 480, 640)
    assert size == 0 and len(vn) == 0
    c.close()


def test_constant_image_no_crash(ctx, oracle):
    """Constant images tie everywhere under the equality-based NMS (SURVEY 8d warns): still must agree with the oracle."""
    img = np.full((1, 64, 96), 128, np.uint8)
    n, kxy, score, desc = ctx.extract(img, kmax=64)
    r = oracle.superpoint(Wt.make_superpoint(seed=7), img[0], kmax=64)
    assert n[0] == r["n"] and np.array_equal(kxy[0], r["kxy"]) and np.array_equal(score[0], r["score"])


def _select_reference(m, kmax, thr, always):
    """numpy statement of the selection rule (oracle: rfe_oracle.c top-k part; published top_k_keypoints)"""
    idx = np.flatnonzero(m.ravel() > thr)
    sc = m.ravel()[idx]
    if idx.size > kmax or always:
        order = np.lexsort((idx, -sc.astype(np.float64)))[:kmax]      # score descending, pixel index ascending
        idx, sc = idx[order], sc[order]
    n = idx.size
    kxy = np.zeros((kmax, 2), np.int32); score = np.zeros((kmax,), np.float32)
    kxy[:n, 0] = idx % m.shape[1]; kxy[:n, 1] = idx // m.shape[1]; score[:n] = sc
    return n, kxy, score


@pytest.mark.parametrize("B,form", [(1, "ordered"), (2, "ordered"), (5, "ordered"), (1, "keys"), (3, "keys"), (4, "keys")])
@pytest.mark.parametrize("density,levels,kmax,always", [(0.5, 7, 1024, 0), (0.9, 3, 4096, 0), (0.02, 5, 1024, 0), (0.02, 5, 1024, 1), (0.6, 0, 333, 0)])
def test_select_kernels_any_candidate_count(ctx, B, form, density, levels, kmax, always):
    """The selection stage alone on synthetic post-NMS maps (rfe_k_select): far more candidates than one LDS window of the rank-all form
    holds (B <= 4; > 16 384 per frame, which only tie-rich maps produce) and than the radix-select form's Kmax (B = 5), with only a few
    distinct score levels so that the cut falls INSIDE a run of equal scores and is decided by the pixel index; also counts below Kmax
    (row-major order, or score order when the graph's TopK is unconditional).  form = "keys": the same maps through rfe_k_select_keys -- the unordered
    64-bit key list of the forward's fused detector tail (one to four frames), ranked by select_rankall_keys_kernel, which must also leave the
    candidate counters at zero for the next call (the hook checks that)."""
    from rover_slam_amd import capi
    H, W = 160, 232
    rng = np.random.default_rng(B * 1000 + kmax + int(density * 100) + always)
    m = rng.uniform(0.001, 1.0, (B, H, W)).astype(np.float32)
    if levels:
        m = (np.floor(m * levels) / levels + 0.01).astype(np.float32)
    m[rng.uniform(size=m.shape) > density] = 0.0
    m[:, :4] = -1; m[:, -4:] = -1; m[:, :, :4] = -1; m[:, :, -4:] = -1
    if density >= 0.5:
        assert int((m[0] > 0.0005).sum()) > 16384
    dm = _dev(ctx, m)
    dn, dk, ds = ctx.alloc(B * 4), ctx.alloc(B * kmax * 8), ctx.alloc(B * kmax * 4)
    fn = capi.lib.rfe_k_select if form == "ordered" else capi.lib.rfe_k_select_keys
    ctx._chk(fn(ctx.h, dm.ptr, B, H, W, kmax, 0.0005, always, dn.ptr, dk.ptr, ds.ptr))
    n = dn.download((B,), np.int32); kxy = dk.download((B, kmax, 2), np.int32); score = ds.download((B, kmax), np.float32)
    for i in range(B):
        rn, rk, rs = _select_reference(m[i], kmax, 0.0005, always)
        assert n[i] == rn
        assert np.array_equal(kxy[i], rk) and np.array_equal(score[i], rs)
    for d in (dm, dn, dk, ds):
        d.free()


@pytest.mark.parametrize("B,H,W,K", [(5, 120, 160, 128), (3, 240, 320, 512), (7, 240, 320, 384), (6, 208, 232, 32), (3, 200, 152, 48)])
def test_stream_mode_equals_pairwise(ctx, oracle, B, H, W, K):
    """rfe_extract_match_stream_dev (B frames, matches (i,i+1), first self block shared per frame) gives the same
    features as rfe_extract_u8 and the same matches as one rfe_match call per pair -- and as the oracle.
    The per-frame self block runs on B sequences: odd B with several query blocks (K >= 256) makes 4*B (sequence, head)
    units that are not a multiple of the 8 XCDs (regression: the attention block decode skipped part of the last frame);
    K < 64 (P+1)/P once overflowed the per-frame rotary tables, which lived in the [P, L, L] similarity buffer."""
    from rover_slam_amd import capi
    frames, _ = synth.make_frames(B, H, W, seed=11)
    dimg = _dev(ctx, frames)
    dn, dk, ds, dd = ctx.alloc(B * 4), ctx.alloc(B * K * 8), ctx.alloc(B * K * 4), ctx.alloc(B * K * 1024)
    dS, dp, dm = ctx.alloc((B - 1) * 4), ctx.alloc((B - 1) * K * 8), ctx.alloc((B - 1) * K * 4)
    ctx._chk(capi.lib.rfe_extract_match_stream_dev(ctx.h, dimg.ptr, H, W, W, B, K, 0.0005, 0.1, dn.ptr, dk.ptr, ds.ptr, dd.ptr,
                                                   dS.ptr, dp.ptr, dm.ptr))
    ctx.synchronize()
    n = dn.download((B,), np.int32); kxy = dk.download((B, K, 2), np.int32); desc = dd.download((B, K, 256), np.float32)
    S = dS.download((B - 1,), np.int32); pairs = dp.download((B - 1, K, 2), np.int32); ms = dm.download((B - 1, K), np.float32)
    n2, kxy2, _, desc2 = ctx.extract(frames, kmax=K)
    assert np.array_equal(n, n2) and np.array_equal(kxy, kxy2) and np.array_equal(desc, desc2)
    wlg = Wt.make_lightglue(seed=11)
    for i in range(B - 1):
        k0 = oracle.normalize_keypoints(kxy[i].astype(np.float32), H, W)
        k1 = oracle.normalize_keypoints(kxy[i + 1].astype(np.float32), H, W)
        S1, p1, m1 = ctx.match(k0[None], k1[None], desc[i][None], desc[i + 1][None], [n[i]], [n[i + 1]])
        assert S[i] == S1[0] and np.array_equal(pairs[i, :S[i]], p1[0, :S1[0]])
        assert np.abs(ms[i, :S[i]] - m1[0, :S1[0]]).max() < (1e-5 if K <= 128 else 5e-4) if S[i] else True   # K > 128: split-key attention in the single-pair call
        r = oracle.lightglue(wlg, k0[:n[i]], k1[:n[i + 1]], desc[i, :n[i]], desc[i + 1, :n[i + 1]])
        assert S[i] == r["S"] and np.array_equal(pairs[i, :S[i]], r["pairs"])
    for d in (dimg, dn, dk, ds, dd, dS, dp, dm):
        d.free()


def test_full_size_640x480_vs_oracle(ctx, oracle):
    """BASELINE size, directly against the oracle: one 640x480 frame pair through SuperPoint (bit-exact) and
    LightGlue at K = 1024 (match list identical, scores within 1e-4)."""
    frames, _ = synth.make_frames(2, 480, 640, seed=640)
    n, kxy, score, desc = ctx.extract(frames, kmax=1024)
    w = Wt.make_superpoint(seed=7)
    for i in range(2):
        r = oracle.superpoint(w, frames[i], kmax=1024)
        assert n[i] == r["n"] == 1024
        assert np.array_equal(kxy[i], r["kxy"]) and np.array_equal(score[i], r["score"]) and np.array_equal(desc[i], r["desc"])
    k0 = oracle.normalize_keypoints(kxy[0].astype(np.float32), 480, 640)
    k1 = oracle.normalize_keypoints(kxy[1].astype(np.float32), 480, 640)
    S, pairs, ms = ctx.match(k0[None], k1[None], desc[0][None], desc[1][None], [1024], [1024])
    r = oracle.lightglue(Wt.make_lightglue(seed=11), k0, k1, desc[0], desc[1])
    assert S[0] == r["S"] and np.array_equal(pairs[0, :S[0]], r["pairs"])        # match assignments identical
    if S[0]:
        assert np.abs(ms[0, :S[0]] - r["ms"]).max() < LG_SCORE_TOL     # stated fp32 tolerance: tests/tolerances.py


def test_stream_b33_640x480_vs_oracle(ctx, oracle):
    """configs[3]'s per-GPU shard exactly as bench.py runs it -- B = 33 frames 640x480, Kmax = 1024, 32 pairs through
    rfe_extract_match_stream_dev -- against the oracle for the first, a middle and the last pair and their frames."""
    from rover_slam_amd import capi
    B, H, W, K = 33, 480, 640, 1024
    frames, _ = synth.make_frames(B, H, W, seed=20240314)
    dimg = _dev(ctx, frames)
    dn, dk, ds, dd = ctx.alloc(B * 4), ctx.alloc(B * K * 8), ctx.alloc(B * K * 4), ctx.alloc(B * K * 1024)
    dS, dp, dm = ctx.alloc((B - 1) * 4), ctx.alloc((B - 1) * K * 8), ctx.alloc((B - 1) * K * 4)
    ctx._chk(capi.lib.rfe_extract_match_stream_dev(ctx.h, dimg.ptr, H, W, W, B, K, 0.0005, 0.1, dn.ptr, dk.ptr, ds.ptr, dd.ptr,
                                                   dS.ptr, dp.ptr, dm.ptr))
    ctx.synchronize()
    n, kxy = dn.download((B,), np.int32), dk.download((B, K, 2), np.int32)
    score, desc = ds.download((B, K), np.float32), dd.download((B, K, 256), np.float32)
    S, pairs, ms = dS.download((B - 1,), np.int32), dp.download((B - 1, K, 2), np.int32), dm.download((B - 1, K), np.float32)
    wsp, wlg = Wt.make_superpoint(seed=7), Wt.make_lightglue(seed=11)
    for i in (0, 15, 31):
        f = [oracle.superpoint(wsp, frames[j], kmax=K) for j in (i, i + 1)]
        for j, r in zip((i, i + 1), f):
            assert n[j] == r["n"] and np.array_equal(kxy[j], r["kxy"]) and np.array_equal(score[j], r["score"]) and np.array_equal(desc[j], r["desc"])
        kn = [oracle.normalize_keypoints(r["kxy"][:r["n"]].astype(np.float32), H, W) for r in f]
        lg = oracle.lightglue(wlg, kn[0], kn[1], f[0]["desc"][:f[0]["n"]], f[1]["desc"][:f[1]["n"]])
        ok, dev = lists_agree(pairs[i, :S[i]], ms[i, :S[i]], lg["pairs"], lg["ms"])
        assert ok and dev < LG_SCORE_TOL, (i, ok, dev)
    for d in (dimg, dn, dk, ds, dd, dS, dp, dm):
        d.free()


def test_lightglue_permutation_equivariance(ctx):
    """Property: permuting the keypoints of one image permutes the matches and nothing else."""
    rng = np.random.default_rng(8)
    n = 200
    d0 = rng.standard_normal((n, 256)).astype(np.float32); d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    perm0 = rng.permutation(n)
    d1 = d0[perm0] + 0.05 * rng.standard_normal((n, 256)).astype(np.float32)
    d1 = (d1 / np.linalg.norm(d1, axis=1, keepdims=True)).astype(np.float32)
    k0 = rng.uniform(-0.9, 0.9, (n, 2)).astype(np.float32)
    k1 = (k0[perm0] + 0.02 * rng.standard_normal((n, 2))).astype(np.float32)
    S, pairs, ms = ctx.match(k0[None], k1[None], d0[None], d1[None], [n], [n])
    base = {(int(i), int(j)): float(s) for (i, j), s in zip(pairs[0, :S[0]], ms[0, :S[0]])}
    q = rng.permutation(n)                       # shuffle image 1's keypoints
    S2, pairs2, ms2 = ctx.match(k0[None], k1[q][None], d0[None], d1[q][None], [n], [n])
    got = {(int(i), int(q[j])): float(s) for (i, j), s in zip(pairs2[0, :S2[0]], ms2[0, :S2[0]])}
    assert len(base) > 20 and set(base) == set(got)
    assert max(abs(base[k] - got[k]) for k in base) < 5e-4   # same stated score tolerance as against the oracle (reduction orders change with the permutation)
    assert (np.diff(pairs2[0, :S2[0], 0]) > 0).all()   # output stays sorted by the index in image 0


@pytest.mark.parametrize("sp_seed,lg_seed,dustbin,batch", [(21, 5, 0.0, 1), (9, 13, 10.5, 7)])
def test_full_size_other_weights_vs_oracle(oracle, sp_seed, lg_seed, dustbin, batch):
    """640x480 against the oracle with other weight sets: a saturated one (K = Kmax) and one with a dustbin bias
    (K < Kmax, different per frame -> ragged LightGlue lengths), extracted alone (latency tilings) and inside a batch
    (throughput tilings).  SuperPoint bit-exact; LightGlue match lists identical, scores within the stated 5e-4."""
    from rover_slam_amd import capi
    c = capi.Context(0)
    wsp = Wt.make_superpoint(seed=sp_seed, dustbin_bias=dustbin)
    wlg = Wt.make_lightglue(seed=lg_seed)
    c.set_weights(capi.KIND_SUPERPOINT, wsp); c.set_weights(capi.KIND_LIGHTGLUE, wlg)
    frames, _ = synth.make_frames(batch + 1, 480, 640, seed=1000 + sp_seed)
    n, kxy, score, desc = c.extract(frames, kmax=1024)
    refs = []
    for i in (0, 1):
        r = oracle.superpoint(wsp, frames[i], kmax=1024)
        assert n[i] == r["n"] and (n[i] < 1024) == (dustbin > 0)
        assert np.array_equal(kxy[i], r["kxy"]) and np.array_equal(score[i], r["score"]) and np.array_equal(desc[i], r["desc"])
        refs.append(r)
    k0 = oracle.normalize_keypoints(kxy[0, :n[0]].astype(np.float32), 480, 640)
    k1 = oracle.normalize_keypoints(kxy[1, :n[1]].astype(np.float32), 480, 640)
    size, vn = c.match_fused(kxy[0, :n[0]].astype(np.float32), kxy[1, :n[1]].astype(np.float32), desc[0, :n[0]], desc[1, :n[1]], 480, 640)
    r = oracle.lightglue(wlg, k0, k1, desc[0, :n[0]], desc[1, :n[1]])
    size_ref, vn_ref = oracle.postprocess_fused(r["pairs"], r["ms"], 0.0, int(n[0]))
    assert size == size_ref and np.array_equal(vn, vn_ref)
    c.close()


def test_stream_mode_repeatable_full_size(ctx):
    """The bench configuration in small: 640x480, B = 5 frames (20 (sequence, head) units, 8 query blocks each) twice into
    fresh buffers -- every output byte identical (a block-decode bug once left half of the last frame's first self block
    uncomputed, which showed as run-to-run differences in the last pair and out-of-bounds reads)."""
    from rover_slam_amd import capi
    B, H, W, K = 5, 480, 640, 1024
    frames, _ = synth.make_frames(B, H, W, seed=4242)
    dimg = _dev(ctx, frames)
    outs = []
    for rep in range(2):
        d = [ctx.alloc(B * 4), ctx.alloc(B * K * 8), ctx.alloc(B * K * 4), ctx.alloc(B * K * 1024),
             ctx.alloc((B - 1) * 4), ctx.alloc((B - 1) * K * 8), ctx.alloc((B - 1) * K * 4)]
        for x, nbytes in zip(d, (B * 4, B * K * 8, B * K * 4, B * K * 1024, (B - 1) * 4, (B - 1) * K * 8, (B - 1) * K * 4)):
            x.upload(np.zeros(nbytes, np.uint8))
        ctx._chk(capi.lib.rfe_extract_match_stream_dev(ctx.h, dimg.ptr, H, W, W, B, K, 0.0005, 0.1, *[x.ptr for x in d]))
        ctx.synchronize()
        outs.append([x.download((nb,), np.uint8) for x, nb in zip(d, (B * 4, B * K * 8, B * K * 4, B * K * 1024, (B - 1) * 4, (B - 1) * K * 8, (B - 1) * K * 4))])
        for x in d:
            x.free()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    # and the last pair agrees with a stand-alone match of the same two frames
    n = outs[0][0].view(np.int32); kxy = outs[0][1].view(np.int32).reshape(B, K, 2); desc = outs[0][3].view(np.float32).reshape(B, K, 256)
    S = outs[0][4].view(np.int32); pairs = outs[0][5].view(np.int32).reshape(B - 1, K, 2)
    i = B - 2
    size, vn = ctx.match_fused(kxy[i, :n[i]].astype(np.float32), kxy[i + 1, :n[i + 1]].astype(np.float32), desc[i, :n[i]], desc[i + 1, :n[i + 1]], H, W)
    want = np.full(n[i], -1, np.int32); want[pairs[i, :S[i], 0]] = pairs[i, :S[i], 1]
    assert size == S[i] and np.array_equal(vn, want)
    dimg.free()


def test_contexts_share_device_weights(oracle):
    """SURVEY 8(b) threading row: several ctxs on one device share the read-only weights.  Same blob -> same device copy
    (rfe_weights_id), other blob -> other copy; a copy outlives the ctx that uploaded it while someone still uses it."""
    from rover_slam_amd import capi
    w7, w9 = Wt.make_superpoint(seed=7), Wt.make_superpoint(seed=9)
    a, b, c3 = capi.Context(0), capi.Context(0), capi.Context(0)
    a.set_weights(capi.KIND_SUPERPOINT, w7); b.set_weights(capi.KIND_SUPERPOINT, w7.copy()); c3.set_weights(capi.KIND_SUPERPOINT, w9)
    ida, idb, idc = (capi.lib.rfe_weights_id(x.h, capi.KIND_SUPERPOINT) for x in (a, b, c3))
    assert ida != 0 and ida == idb and idc != ida
    assert capi.lib.rfe_weights_id(a.h, capi.KIND_LIGHTGLUE) == 0
    a.close()                                                   # b keeps the shared copy alive
    img = synth.make_frames(1, 120, 160, seed=12)[0][0]
    n, kxy, score, desc = b.extract(img, kmax=200)
    r = oracle.superpoint(w7, img, kmax=200)
    assert n[0] == r["n"] and np.array_equal(kxy[0], r["kxy"]) and np.array_equal(desc[0], r["desc"])
    b.set_weights(capi.KIND_SUPERPOINT, w9)                     # switching blobs: now shares c3's copy
    assert capi.lib.rfe_weights_id(b.h, capi.KIND_SUPERPOINT) == idc
    n, kxy, score, desc = b.extract(img, kmax=200)
    r = oracle.superpoint(w9, img, kmax=200)
    assert n[0] == r["n"] and np.array_equal(kxy[0], r["kxy"]) and np.array_equal(desc[0], r["desc"])
    b.close(); c3.close()
