#!/usr/bin/env python3
"""Experiment (not product code): does running two independent half batches on two HIP streams (two contexts) hide the
launch/drain tails of the kernels?  Compares 2 x (17 frames + 16 pairs) sequentially on one stream against the same
two half batches concurrently on two streams.  usage: python tools/exp_two_streams.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rover_slam_amd import capi, weights as Wt, synth

H, W, K, B = 480, 640, 1024, int(os.environ.get("EXP_B", "17"))
dev = torch.device("cuda", 0)
wsp, wlg = Wt.make_superpoint(seed=7), Wt.make_lightglue(seed=11)
frames = torch.from_numpy(synth.make_frames(2 * B - 1, H, W)[0]).to(dev)
halves = [frames[:B].contiguous(), frames[B - 1:].contiguous()]


def make(stream):
    c = capi.Context(0)
    c.set_weights(capi.KIND_SUPERPOINT, wsp); c.set_weights(capi.KIND_LIGHTGLUE, wlg)
    c.set_stream(stream.cuda_stream)
    bufs = dict(n=torch.zeros(B, dtype=torch.int32, device=dev), kxy=torch.zeros(B, K, 2, dtype=torch.int32, device=dev),
                score=torch.zeros(B, K, device=dev), desc=torch.zeros(B, K, 256, device=dev),
                S=torch.zeros(B - 1, dtype=torch.int32, device=dev), pairs=torch.zeros(B - 1, K, 2, dtype=torch.int32, device=dev),
                ms=torch.zeros(B - 1, K, device=dev))
    return c, bufs


def run(c, b, img):
    c._chk(capi.lib.rfe_extract_match_stream_dev(c.h, img.data_ptr(), H, W, W, B, K, 0.0005, 0.1, b["n"].data_ptr(), b["kxy"].data_ptr(),
                                                 b["score"].data_ptr(), b["desc"].data_ptr(), b["S"].data_ptr(), b["pairs"].data_ptr(),
                                                 b["ms"].data_ptr()))


s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
c1, b1 = make(s1)
c2, b2 = make(s2)
c3, b3 = make(s1)      # second context on the SAME stream: the sequential baseline
for mode in ("sequential", "concurrent", "sequential", "concurrent"):
    ca, ba, cb, bb = (c1, b1, c3, b3) if mode == "sequential" else (c1, b1, c2, b2)
    for _ in range(3):
        run(ca, ba, halves[0]); run(cb, bb, halves[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 10
    for _ in range(steps):
        run(ca, ba, halves[0]); run(cb, bb, halves[1])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{mode}: {dt * 1e3:.3f} ms per 2 x ({B} frames + {B - 1} pairs) -> {2 * (B - 1) / dt:.1f} frames/s")
