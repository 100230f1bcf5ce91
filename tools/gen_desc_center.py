#!/usr/bin/env python3
"""Writes rover-slam_amd/data/sp_desc_center_seed<S>.npy: the mean of convDb's pre-bias output over the cells of one calibration
frame, for `weights.make_superpoint(seed=S, desc_center="auto")` (an LSUV-style data-dependent initialisation of the synthetic
descriptor head; see that docstring for why).  Build-container tool: uses the CPU oracle's conv (test infrastructure) for the
encoder and float64 numpy for the 1x1 head; the output is 256 floats of DATA, committed.

    python tools/gen_desc_center.py --seed 7
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--frame-seed", type=int, default=7)
    a = ap.parse_args(argv)
    from rover_slam_amd import weights as Wt, synth
    from oracle import oracle as O
    O.build()
    w = Wt.make_superpoint(seed=a.seed)
    man = {n: (o, s) for n, o, s in Wt.sp_manifest()[0]}
    get = lambda n: w[man[n][0]:man[n][0] + int(np.prod(man[n][1]))].reshape(man[n][1])
    img = synth.make_frames(1, 240, 320, seed=a.frame_seed)[0][0]
    feat = O.superpoint(w, img, kmax=16, debug=True)["feat"]                       # conv4b output [Hc, Wc, 128]
    h = O.conv3x3(feat, get("convDa.weight"), get("convDa.bias"), relu=True)       # [Hc, Wc, 256]
    raw = h.reshape(-1, 256).astype(np.float64) @ get("convDb.weight").reshape(256, 256).astype(np.float64).T
    center = raw.mean(0).astype(np.float32)
    out = os.path.join(ROOT, "rover-slam_amd", "data", f"sp_desc_center_seed{a.seed}.npy")
    np.save(out, center)
    print(f"{out}: |centre| = {np.linalg.norm(center):.3f}, per-channel std of the head output = {raw.std(0).mean():.3f}")


if __name__ == "__main__":
    main()
