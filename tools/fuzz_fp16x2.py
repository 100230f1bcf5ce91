#!/usr/bin/env python3
"""Randomised sweep of RFE_OPT_LG_FP16X2 against the oracle (GPU box; test infrastructure, not product code): batches of 16-20 pairs
(>= 32 768 token rows: the split GEMMs and the split attention are selected) with random ragged keypoint counts, random LightGlue weight
seeds and a padded length that is any multiple of 4 up to 1024; every pair against oracle.lightglue under the tests' borderline rule and
the stated score tolerance, and against the fp32 path of the same call.  usage: python tools/fuzz_fp16x2.py [seconds=300] [seed=0]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
from rover_slam_amd import capi, weights as Wt
from tolerances import LG_SCORE_TOL, lists_agree_borderline
from test_gpu_throughput_parity import _constructed_batch


def main(seconds=300.0, seed=0):
    rng = np.random.default_rng(seed)
    O.build()
    c = capi.Context(0)
    t0, it, fails, pairs_checked, worst, worst32 = time.time(), 0, 0, 0, 0.0, 0.0
    while time.time() - t0 < seconds:
        it += 1
        P = int(rng.integers(16, 21))
        K = int(rng.choice([1024, 1024, 1020, 1000, 964]))
        wseed = int(rng.integers(11, 40))
        wlg = Wt.make_lightglue(seed=wseed)
        c.set_weights(capi.KIND_LIGHTGLUE, wlg)
        lens0 = [int(rng.integers(32, K + 1)) if rng.random() < 0.5 else K for _ in range(P)]
        lens1 = [int(rng.integers(32, K + 1)) if rng.random() < 0.5 else K for _ in range(P)]
        if 2 * P * K < 32768:
            continue
        k0, k1, d0, d1 = _constructed_batch(P, K, int(rng.integers(1 << 30)), lens0, lens1)
        fold = int(rng.integers(0, 2))
        c.set_option(capi.OPT_LG_FOLD_WO, fold)
        S32, p32, m32 = c.match(k0, k1, d0, d1, lens0, lens1)
        c.set_option(capi.OPT_LG_FP16X2, 1)
        try:
            S, pairs, ms = c.match(k0, k1, d0, d1, lens0, lens1)
        finally:
            c.set_option(capi.OPT_LG_FP16X2, 0)
        check = rng.choice(P, 4, replace=False)          # the oracle takes ~1 s per full-size pair: four pairs per batch
        for p in check:
            m, n = lens0[p], lens1[p]
            ref = O.lightglue(wlg, k0[p, :m], k1[p, :n], d0[p, :m], d1[p, :n], debug=True)
            ok, dev, only = lists_agree_borderline(pairs[p, :S[p]], ms[p, :S[p]], ref["pairs"], ref["ms"], ref["scores"], K)
            ok32, dev32, only32 = lists_agree_borderline(p32[p, :S32[p]], m32[p, :S32[p]], ref["pairs"], ref["ms"], ref["scores"], K)
            pairs_checked += 1
            worst, worst32 = max(worst, dev), max(worst32, dev32)
            if not (ok and dev < LG_SCORE_TOL and ok32 and dev32 < LG_SCORE_TOL):
                fails += 1
                print(f"FAIL it {it} pair {p}: P={P} K={K} lens=({m},{n}) wseed={wseed} fold={fold}: fp16x2 ok={ok} dev={dev:.2e} one-sided={only} | fp32 ok={ok32} dev={dev32:.2e} one-sided={only32}", flush=True)
        print(f"it {it}: P={P} K={K} wseed={wseed} fold={fold} matches/pair {int(S.mean())}  worst dev so far fp16x2 {worst:.2e} fp32 {worst32:.2e}", flush=True)
    c.close()
    print(f"fuzz_fp16x2: {it} batches, {pairs_checked} pairs checked against the oracle, {fails} failures; max |score dev| vs oracle: fp16x2 {worst:.2e}, fp32 path {worst32:.2e} (tolerance {LG_SCORE_TOL:.0e})")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
