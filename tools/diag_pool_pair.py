#!/usr/bin/env python3
"""Diagnostic (GPU box): tests/test_pool.py's 11-frame stream at Kmax 256 -- every pair of the whole-stream call and of the 4- / 3-pair
shard calls against the CPU oracle and a float64 evaluation: where do two fp32 tilings sit relative to each other and to the references?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from rover_slam_amd import capi, weights as Wt, synth
from oracle import oracle as O
from lg_tolerance_study import lg_f64, score_dev
import test_pool as tp
O.build()
F, kmax, H, W = 11, 256, 240, 320
frames, _ = synth.make_frames(F, H, W, seed=9)
wsp, wlg = Wt.make_superpoint(seed=7), Wt.make_lightglue(seed=11)
ctx = capi.Context(0)
ctx.set_weights(capi.KIND_SUPERPOINT, wsp); ctx.set_weights(capi.KIND_LIGHTGLUE, wlg)
whole = tp._single_ctx_stream(ctx, frames, kmax)
parts = {}
for r in range(3):
    first, nfr, own = capi.pool_shard(F, 3, r)
    part = tp._single_ctx_stream(ctx, np.ascontiguousarray(frames[first:first + nfr]), kmax)
    for p in range(own):
        parts[first + p] = (part["pairs"][p, :part["S"][p]], part["ms"][p, :part["S"][p]])
print("| pair | S | whole vs shard | whole vs oracle | shard vs oracle | whole vs f64 | shard vs f64 | oracle vs f64 |")
print("|---:|---:|---:|---:|---:|---:|---:|---:|")
n, kxy, desc = whole["n"], whole["kxy"], whole["desc"]
for p in range(F - 1):
    k0 = O.normalize_keypoints(kxy[p, :n[p]].astype(np.float32), H, W); k1 = O.normalize_keypoints(kxy[p + 1, :n[p + 1]].astype(np.float32), H, W)
    ref = O.lightglue(wlg, k0, k1, desc[p, :n[p]], desc[p + 1, :n[p + 1]])
    p64, m64, _ = lg_f64(wlg, k0, k1, desc[p, :n[p]], desc[p + 1, :n[p + 1]])
    w = (whole["pairs"][p, :whole["S"][p]], whole["ms"][p, :whole["S"][p]]); s = parts[p]
    d = lambda a, b: score_dev(a[0], a[1], b[0], b[1])[1]
    o = (ref["pairs"], ref["ms"]); f = (p64, m64)
    print(f"| {p} | {int(whole['S'][p])} | {d(w, s):.2e} | {d(w, o):.2e} | {d(s, o):.2e} | {d(w, f):.2e} | {d(s, f):.2e} | {d(o, f):.2e} |")
ctx.close()
