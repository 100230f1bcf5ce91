"""Synthetic 8-bit grayscale frames for tests and bench: Gaussian blobs +
axis-aligned rectangles + low-amplitude noise, consecutive frames related by a small integer
translation so that LightGlue sees real correspondences.  Never constant images (the NMS of the
reference graph is equality based; constant regions tie everywhere)."""
import numpy as np


def make_scene(rng, H, W, margin=16):
    Hs, Ws = H + 2 * margin, W + 2 * margin
    yy, xx = np.mgrid[0:Hs, 0:Ws].astype(np.float32)
    img = np.zeros((Hs, Ws), np.float32)
    for _ in range(64):
        cx, cy = rng.uniform(0, Ws), rng.uniform(0, Hs)
        s = rng.uniform(2.0, 12.0)
        a = rng.uniform(40.0, 200.0)
        img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    for _ in range(32):
        x0, y0 = int(rng.integers(0, Ws - 8)), int(rng.integers(0, Hs - 8))
        w, h = int(rng.integers(6, 64)), int(rng.integers(6, 64))
        img[y0:y0 + h, x0:x0 + w] += rng.uniform(-60.0, 60.0)
    return img


def make_frames(n, H=480, W=640, seed=20240314, max_shift=8, shift_step=1):
    """Returns uint8 [n,H,W] and the per-frame integer (dx,dy) offsets into the scene.
    shift_step = 8 (with max_shift = 16): consecutive frames differ by multiples of SuperPoint's 8-px cell, so that the same scene
    point lands on the same place of a cell and even an untrained (seeded) extractor repeats its keypoints and descriptors --
    what gives the bench's self-check hundreds of matches per pair.  shift_step = 1 is the stream of rounds 1-4, bit for bit."""
    rng = np.random.default_rng(seed)
    assert max_shift % shift_step == 0
    margin = 2 * max_shift
    scene = make_scene(rng, H, W, margin)
    frames = np.empty((n, H, W), np.uint8)
    offs = np.zeros((n, 2), np.int32)
    ox = oy = margin
    for i in range(n):
        if i:
            ox = int(np.clip(ox + shift_step * rng.integers(-(max_shift // shift_step), max_shift // shift_step + 1), 0, 2 * margin))
            oy = int(np.clip(oy + shift_step * rng.integers(-(max_shift // shift_step), max_shift // shift_step + 1), 0, 2 * margin))
        offs[i] = (ox, oy)
        f = scene[oy:oy + H, ox:ox + W] + rng.integers(0, 16, (H, W)).astype(np.float32)
        frames[i] = np.clip(f, 0, 255).astype(np.uint8)
    return frames, offs
