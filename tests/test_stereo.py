"""Sparse stereo matching: oracle restatement of Frame::ComputeStereoMatches
(reference src/Frame.cc:1159-1446) on a constructed case with a known answer (CPU), and the HIP kernels
against the oracle on extracted features (GPU)."""
import numpy as np
import pytest

from rover_slam_amd import weights as Wt, synth


def _shifted_pair(H, W, disp, seed):
    """left / right views of one textured scene related by a pure horizontal shift of `disp` px."""
    rng = np.random.default_rng(seed)
    scene = synth.make_scene(rng, H, W + disp, margin=0)
    # independent sensor noise per view (identical views give SAD = 0 everywhere, whose zero median makes the
    # reference's outlier cut discard every match)
    # left(x) = scene(x), right(x) = scene(x + disp): a feature at uL appears at uR = uL - disp
    left = np.clip(scene[:, :W] + rng.integers(0, 8, (H, W)), 0, 255).astype(np.uint8)
    right = np.clip(scene[:, disp:disp + W] + rng.integers(0, 8, (H, W)), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(left), np.ascontiguousarray(right)


def test_oracle_stereo_known_disparity(oracle):
    H, W, disp = 64, 160, 7
    rng = np.random.default_rng(0)
    right = rng.integers(0, 256, (H, W)).astype(np.uint8)
    left = np.roll(right, disp, axis=1)                     # left(x) = right(x - disp): feature at uL maps to uL - disp
    left = np.clip(left.astype(np.int32) + rng.integers(-2, 3, left.shape), 0, 255).astype(np.uint8)   # SAD > 0: a zero
    # median would make the reference's outlier cut (SAD >= 2.1 * median) discard every match
    kl = np.array([[60, 20], [100, 40], [30, 30], [5, 30]], np.float32)     # last one: patch leaves the image -> skipped
    kr = np.array([[93, 41], [53, 20], [23, 33], [120, 20]], np.float32)    # rows within +-2; (23,33) is 3 rows off
    d = rng.standard_normal((4, 256)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dl = d.copy()
    dr = np.stack([d[1], d[0], d[2], -d[0]])                # right 0 <-> left 1, right 1 <-> left 0
    mb, mbf = 0.11, 0.11 * 435.0
    u, z = oracle.stereo_match(left, right, kl, kr, dl, dr, mb, mbf)
    assert np.abs(u[:2] - [60 - disp, 100 - disp]).max() < 0.05 and np.abs(z[:2] - mbf / disp).max() < 0.05
    assert u[2] == -1 and z[2] == -1 and u[3] == -1        # row band violated / patch outside the image
    # a disparity beyond mbf/mb is rejected
    u2, _ = oracle.stereo_match(left, right, kl, kr, dl, dr, 10.0, 10.0 * 0.5)
    assert (u2 == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,disp", [(120, 160, 9), (240, 320, 17)])
def test_stereo_gpu_vs_oracle(oracle, H, W, disp):
    from rover_slam_amd import capi
    ctx = capi.Context(0)
    ctx.set_weights(capi.KIND_SUPERPOINT, Wt.make_superpoint(seed=7))
    left, right = _shifted_pair(H, W, disp, seed=disp)
    n, kxy, score, desc = ctx.extract(np.stack([left, right]), kmax=400)
    kl, kr = kxy[0, :n[0]].astype(np.float32), kxy[1, :n[1]].astype(np.float32)
    dl, dr = desc[0, :n[0]], desc[1, :n[1]]
    mb, mbf = 0.11, 0.11 * 435.0
    u, z = ctx.stereo_match(left, right, kl, kr, dl, dr, mb, mbf)
    u_ref, z_ref = oracle.stereo_match(left, right, kl, kr, dl, dr, mb, mbf)
    assert np.array_equal(u, u_ref) and np.array_equal(z, z_ref)          # exact: integer SAD + IEEE float ops
    got = u >= 0
    assert got.sum() > 10                                                  # the test is not vacuous
    err = np.abs((kl[got, 0] - u[got]) - disp)                             # recovered disparity vs the true shift
    assert np.median(err) < 0.5 and (err < 1.0).mean() > 0.8               # (random-weight descriptors: a few coarse mismatches)
    # empty right set / single keypoint
    u0, _ = ctx.stereo_match(left, right, kl, kr[:0], dl, dr[:0], mb, mbf)
    assert (u0 == -1).all()
    ctx.close()


@pytest.mark.gpu
def test_distance_matrix_and_binarize():
    """SURVEY 8(f) N3 / N4 helpers: all-pairs DescriptorDistance_sp and Frame::binarize_descriptors."""
    from rover_slam_amd import capi
    ctx = capi.Context(0)
    rng = np.random.default_rng(4)
    a = rng.standard_normal((37, 256)).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.standard_normal((101, 256)).astype(np.float32); b /= np.linalg.norm(b, axis=1, keepdims=True)
    out = np.empty((37, 101), np.float32)
    ctx._chk(capi.lib.rfe_l2_distance_matrix(ctx.h, a.ctypes.data, 37, b.ctypes.data, 101, out.ctypes.data))
    ref = np.linalg.norm(a[:, None, :].astype(np.float64) - b[None].astype(np.float64), axis=2)
    assert np.abs(out - ref).max() < 1e-6
    # the thresholds the reference applies to these distances (TH_LOW / TH_HIGH, SPmatcher.cc:13-14)
    assert ((out < 1.2) == (ref < 1.2)).all() and ((out < 1.4) == (ref < 1.4)).all()
    bits = np.empty((37, 256), np.uint8)
    ctx._chk(capi.lib.rfe_binarize_descriptors(ctx.h, a.ctypes.data, 37, bits.ctypes.data))
    assert np.array_equal(bits, (a > 0).astype(np.uint8))
    ctx.close()
