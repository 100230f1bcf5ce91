#!/usr/bin/env python3
"""Concurrency sweep (GPU box; test infrastructure): N host threads, one ctx each on the same device (the reference runs
its left / right extractors and three matchers from different threads, src/Frame.cc:142-147), sharing one device copy of the
weights, each doing random extractions and matches checked against the oracle.
usage: python tools/fuzz_threads.py [seconds=40] [threads=3]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from rover_slam_amd import capi, synth, weights as Wt

wsp, wlg = Wt.make_superpoint(seed=7), Wt.make_lightglue(seed=11)
fails, cases = [], [0]
olock = threading.Lock()          # the oracle's OpenMP pool is shared: one oracle call at a time


def worker(tid, seconds):
    rng = np.random.default_rng(100 + tid)
    c = capi.Context(0)
    c.set_weights(capi.KIND_SUPERPOINT, wsp); c.set_weights(capi.KIND_LIGHTGLUE, wlg)
    t0 = time.time()
    while time.time() - t0 < seconds:
        H, W = 8 * int(rng.integers(8, 31)), 8 * int(rng.integers(8, 41))
        K = int(rng.choice([64, 200, 512]))
        frames, _ = synth.make_frames(2, H, W, seed=int(rng.integers(1 << 30)))
        n, kxy, score, desc = c.extract(frames, kmax=K)
        size, vn = c.match_fused(kxy[0, :n[0]].astype(np.float32), kxy[1, :n[1]].astype(np.float32), desc[0, :n[0]], desc[1, :n[1]], H, W)
        with olock:
            ok = True
            for i in range(2):
                r = O.superpoint(wsp, frames[i], kmax=K)
                ok &= bool(n[i] == r["n"] and np.array_equal(kxy[i], r["kxy"]) and np.array_equal(desc[i], r["desc"]))
            if n[0] and n[1]:
                r = O.lightglue(wlg, O.normalize_keypoints(kxy[0, :n[0]].astype(np.float32), H, W),
                                O.normalize_keypoints(kxy[1, :n[1]].astype(np.float32), H, W), desc[0, :n[0]], desc[1, :n[1]], debug=True)
                sref, vref = O.postprocess_fused(r["pairs"], r["ms"], 0.0, int(n[0]))
                if not (size == sref and np.array_equal(vn, vref)):
                    diff = np.nonzero(vn != vref)[0]
                    sc = r["scores"]
                    border = all(abs(np.exp(sc[i].max()) - 0.1) < 2e-4 or np.sort(sc[i])[-1] - np.sort(sc[i])[-2] < 2e-4 for i in diff)
                    ok &= bool(border)
            cases[0] += 1
            if not ok:
                fails.append(f"thread {tid}: H={H} W={W} K={K}")
                print("MISMATCH", fails[-1], flush=True)
    ids = (capi.lib.rfe_weights_id(c.h, capi.KIND_SUPERPOINT), capi.lib.rfe_weights_id(c.h, capi.KIND_LIGHTGLUE))
    with olock:
        worker.ids.append(ids)
    c.close()


worker.ids = []

if __name__ == "__main__":
    a = sys.argv[1:]
    seconds, nthr = (float(a[0]) if a else 40.0), (int(a[1]) if len(a) > 1 else 3)
    O.build()
    th = [threading.Thread(target=worker, args=(t, seconds)) for t in range(nthr)]
    [t.start() for t in th]
    [t.join() for t in th]
    shared = len(set(worker.ids)) == 1
    print(f"fuzz_threads: {cases[0]} cases on {nthr} threads, {len(fails)} mismatches, weights shared: {shared}")
    sys.exit(1 if fails or not shared else 0)
