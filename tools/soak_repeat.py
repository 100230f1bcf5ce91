#!/usr/bin/env python3
"""Repeatability soak (round 5, after the hoisted-LDS-read race): the same call, thousands of times, must return the same BYTES -- alone and while a second context on
another thread keeps the chip busy with different work (timing perturbation is what exposed the race: it only showed when SuperPoint's two heads overlapped differently).
Shapes: small batches (where kernels overlap most), one frame / one pair / one stereo frame of the latency configurations, the throughput batch.
GPU box:  python tools/soak_repeat.py [seconds per case = 20]  > profiles/rNN_soak.md"""
import hashlib
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rover_slam_amd import capi, synth, weights as Wt  # noqa: E402


def digest(arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def stream_call(ctx, dimg, B, H, W, K, bufs):
    ctx._chk(capi.lib.rfe_extract_match_stream_dev(ctx.h, dimg.ptr, H, W, W, B, K, 0.0005, 0.1, *[b.ptr for b in bufs]))
    ctx.synchronize()
    shapes = [((B,), np.int32), ((B, K, 2), np.int32), ((B, K), np.float32), ((B, K, 256), np.float32), ((max(B - 1, 1),), np.int32),
              ((max(B - 1, 1), K, 2), np.int32), ((max(B - 1, 1), K), np.float32)]
    out = [b.download(s, d) for b, (s, d) in zip(bufs, shapes)]
    n, S = out[0], out[4]
    # match rows are only defined up to S, descriptor rows up to n: hash the defined part
    return digest([n, out[1], out[2]] + [out[3][i, :n[i]] for i in range(B)] + ([S] + [out[5][q, :S[q]] for q in range(B - 1)] + [out[6][q, :S[q]] for q in range(B - 1)] if B > 1 else []))


def main(seconds=20.0):
    wsp, wlg = Wt.make_superpoint(seed=7, desc_center="auto"), Wt.make_lightglue(seed=11, calibrated=True)
    ctx = capi.Context(0)
    ctx.set_weights(capi.KIND_SUPERPOINT, wsp); ctx.set_weights(capi.KIND_LIGHTGLUE, wlg)
    stop = threading.Event()

    def disturb():      # a second context: extraction + matching of other shapes, back to back
        c2 = capi.Context(0)
        c2.set_weights(capi.KIND_SUPERPOINT, wsp); c2.set_weights(capi.KIND_LIGHTGLUE, wlg)
        rng = np.random.default_rng(1)
        fr = [synth.make_frames(int(b), int(h), int(w), seed=3)[0] for b, h, w in ((1, 480, 640), (3, 120, 160), (2, 240, 320), (6, 96, 128))]
        i = 0
        while not stop.is_set():
            f = fr[i % len(fr)]
            n, kxy, sc, de = c2.extract(f, kmax=int(rng.choice([64, 256, 1024])))
            if f.shape[0] > 1 and n[0] > 0 and n[1] > 0:
                c2.match_fused(kxy[0, :n[0]].astype(np.float32), kxy[1, :n[1]].astype(np.float32), de[0, :n[0]], de[1, :n[1]], f.shape[1], f.shape[2])
            i += 1
        c2.close()

    print("# Repeatability soak (tools/soak_repeat.py): identical bytes from every call, alone and next to a second busy context\n")
    print("| case | calls alone | calls with a second context | distinct results |")
    print("|---|---:|---:|---:|")
    bad = 0
    for B, H, W, K in ((5, 120, 160, 128), (3, 200, 152, 48), (4, 120, 160, 300), (1, 480, 640, 1024), (2, 480, 640, 1024), (2, 480, 752, 1024), (33, 480, 640, 1024)):
        frames, _ = synth.make_frames(B, H, W, seed=20240314, max_shift=16, shift_step=8)
        dimg = ctx.alloc(frames.nbytes).upload(frames)
        P = max(B - 1, 1)
        bufs = [ctx.alloc(x) for x in (B * 4, B * K * 8, B * K * 4, B * K * 1024, P * 4, P * K * 8, P * K * 4)]
        seen, counts = set(), []
        for phase in (0, 1):
            th = None
            if phase:
                stop.clear()
                th = threading.Thread(target=disturb); th.start()
                time.sleep(0.5)
            t0, calls = time.time(), 0
            while time.time() - t0 < seconds:
                seen.add(stream_call(ctx, dimg, B, H, W, K, bufs) if B > 1 else digest(ctx.extract(frames, kmax=K)))
                calls += 1
            counts.append(calls)
            if th is not None:
                stop.set(); th.join()
        bad += len(seen) != 1
        print(f"| {B} x {H}x{W}, Kmax {K} | {counts[0]} | {counts[1]} | **{len(seen)}** |", flush=True)
        for b in bufs + [dimg]:
            b.free()
    ctx.close()
    print(f"\n{'ALL REPEATABLE' if not bad else str(bad) + ' CASE(S) NOT REPEATABLE'}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 20.0))
