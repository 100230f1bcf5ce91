#!/usr/bin/env python3
"""Repeatability check: the batched stream call on identical inputs must give identical bytes every time."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rover_slam_amd import capi, weights as Wt, synth

H, W, K, B = [int(v) for v in os.environ.get("CFG", "480,640,1024,33").split(",")]
dev = torch.device("cuda", 0)
c = capi.Context(0)
c.set_weights(capi.KIND_SUPERPOINT, Wt.make_superpoint(seed=7)); c.set_weights(capi.KIND_LIGHTGLUE, Wt.make_lightglue(seed=11))
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st); c.set_stream(st.cuda_stream)
frames = torch.from_numpy(synth.make_frames(B, H, W)[0]).to(dev)
names = ("n", "kxy", "score", "desc", "S", "pairs", "ms")
def bufs():
    return [torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, K, 2, dtype=torch.int32, device=dev), torch.zeros(B, K, device=dev),
            torch.zeros(B, K, 256, device=dev), torch.zeros(B - 1, dtype=torch.int32, device=dev),
            torch.zeros(B - 1, K, 2, dtype=torch.int32, device=dev), torch.zeros(B - 1, K, device=dev)]
def run(b):
    c._chk(capi.lib.rfe_extract_match_stream_dev(c.h, frames.data_ptr(), H, W, W, B, K, 0.0005, 0.1, *[t.data_ptr() for t in b]))
ref = bufs(); run(ref); torch.cuda.synchronize()
bad = 0
for it in range(int(os.environ.get("ITERS", "30"))):
    b = bufs(); run(b); torch.cuda.synchronize()
    for nm, x, y in zip(names, b, ref):
        if not torch.equal(x, y):
            d = (x != y)
            idx = d.nonzero()[:3].tolist()
            print(f"iter {it}: {nm} differs in {int(d.sum())} elements, first at {idx}", flush=True)
            if nm == "pairs":
                p = idx[0][0]
                print("  S", int(b[4][p]), int(ref[4][p]), "rows", x[p, idx[0][1]].tolist(), y[p, idx[0][1]].tolist(),
                      "ms", float(b[6][p, idx[0][1]]), float(ref[6][p, idx[0][1]]))
            bad += 1
print("mismatching outputs:", bad)
