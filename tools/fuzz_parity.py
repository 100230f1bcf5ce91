#!/usr/bin/env python3
"""Randomised shape sweep of the C ABI against the oracle (GPU box; test infrastructure, not product code).
SuperPoint: random H, W (any size >= 48), batch, Kmax, u8 or float entry -> every output bit-exact.  LightGlue: random pair counts and
ragged (m, n) incl. tiny sets -> match lists identical, scores within 1e-4 (small sets).  Stream mode: random B, K.
usage: python tools/fuzz_parity.py [seconds=60] [seed=0]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from rover_slam_amd import capi, synth, weights as Wt
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tolerances import borderline, _top2_gap as top2_gap  # noqa: E402  (the one statement of the borderline rule)


def main(seconds=60.0, seed=0, max_cases=None):
    """max_cases: stop after that many cases whatever the clock says (the pytest short run: the SAME cases on every box, fast or busy)"""
    rng = np.random.default_rng(seed)
    O.build()
    c = capi.Context(0)
    wsp, wlg = Wt.make_superpoint(seed=7), Wt.make_lightglue(seed=11)
    wsp2 = Wt.make_superpoint(seed=9, dustbin_bias=7.0)
    c.set_weights(capi.KIND_LIGHTGLUE, wlg)
    t0, it, fails = time.time(), 0, 0
    while time.time() - t0 < seconds and (max_cases is None or it < max_cases):
        it += 1
        kind = it % 5
        if kind == 0:                                             # ---- SuperPoint
            big = rng.random() < 0.08
            H, W = (480, int(rng.choice([640, 752]))) if big else (8 * int(rng.integers(6, 33)), 8 * int(rng.integers(6, 41)))
            if not big and rng.random() < 0.4:          # any size >= 8 goes (floor pooling): not a multiple of 8 / 4 / 2
                H, W = H + int(rng.integers(0, 8)), W + int(rng.integers(0, 8))
            B, K = int(rng.integers(1, 6)), int(rng.choice([1, 7, 64, 100, 333, 512, 1024, 4096]))
            w = wsp if rng.random() < 0.6 else wsp2
            c.set_weights(capi.KIND_SUPERPOINT, w)
            frames, _ = synth.make_frames(B, H, W, seed=int(rng.integers(1 << 30)))
            pad = int(rng.choice([0, 0, 3, 8, 40]))
            src = frames
            if rng.random() < 0.3:        # the float entry (Extractor_Inference on CV_32F): values off the 1/255 lattice and outside [0, 1]
                pad = -1
                src = (frames.astype(np.float32) * np.float32(rng.uniform(0.002, 0.008)) + np.float32(rng.uniform(-0.3, 0.3))).astype(np.float32)
                n, kxy, score, desc = c.extract_f32(src, kmax=K)
            else:
                n, kxy, score, desc = c.extract(frames, kmax=K, pad_cols=pad)
            ok = True
            for i in range(B):
                r = O.superpoint(w, src[i], kmax=K)
                ok &= bool(n[i] == r["n"] and np.array_equal(kxy[i], r["kxy"]) and np.array_equal(score[i], r["score"]) and np.array_equal(desc[i], r["desc"]))
            tag = f"sp H={H} W={W} B={B} K={K} pad={pad}"
        elif kind == 1:                                           # ---- LightGlue, ragged batch
            P = int(rng.integers(1, 5))
            hi = 1025 if rng.random() < 0.1 else 400
            Mmax, Nmax = int(rng.integers(1, hi)), int(rng.integers(1, hi))
            ms_, ns_ = [int(rng.integers(0, Mmax + 1)) for _ in range(P)], [int(rng.integers(0, Nmax + 1)) for _ in range(P)]
            ms_[0], ns_[0] = Mmax, Nmax
            k0 = rng.uniform(-0.9, 0.9, (P, Mmax, 2)).astype(np.float32); k1 = rng.uniform(-0.9, 0.9, (P, Nmax, 2)).astype(np.float32)
            d0 = rng.standard_normal((P, Mmax, 256)).astype(np.float32); d0 /= np.linalg.norm(d0, axis=2, keepdims=True)
            d1 = rng.standard_normal((P, Nmax, 256)).astype(np.float32); d1 /= np.linalg.norm(d1, axis=2, keepdims=True)
            for p in range(P):                                     # give set 1 real correspondences
                m = min(ms_[p], ns_[p])
                if m:
                    d1[p, :m] = d0[p, :m] + 0.05 * rng.standard_normal((m, 256)).astype(np.float32)
                    d1[p, :m] /= np.linalg.norm(d1[p, :m], axis=1, keepdims=True); k1[p, :m] = k0[p, :m] + 0.01
            h2 = rng.random() < 0.3          # RFE_OPT_LG_FP16X2: the split forms of the one- / few-pair kernels, same bars
            c.set_option(capi.OPT_LG_FP16X2, 1 if h2 else 0)
            try:
                S, pairs, msc = c.match(k0, k1, d0, d1, ms_, ns_)
            finally:
                c.set_option(capi.OPT_LG_FP16X2, 0)
            ok = True
            for p in range(P):
                if ms_[p] == 0 or ns_[p] == 0:
                    ok &= bool(S[p] == 0)
                    continue
                r = O.lightglue(wlg, k0[p, :ms_[p]], k1[p, :ns_[p]], d0[p, :ms_[p]], d1[p, :ns_[p]], debug=True)
                good = bool(S[p] == r["S"] and np.array_equal(pairs[p, :S[p]], r["pairs"]) and (S[p] == 0 or np.abs(msc[p, :S[p]] - r["ms"]).max() < 1e-4))
                if not good:
                    # classify: a match whose score sits within 1e-4 of the 0.1 filter, or an argmax tie within 1e-4, may legitimately flip
                    gp = {tuple(x) for x in pairs[p, :S[p]].tolist()}; rp = {tuple(x) for x in r["pairs"].tolist()}
                    sc = r["scores"]
                    near = all(borderline(sc, i, j, max(ms_[p], ns_[p])) for (i, j) in gp ^ rp)
                    common = sorted(gp & rp)
                    gm = {tuple(x): v for x, v in zip(pairs[p, :S[p]].tolist(), msc[p, :S[p]])}; rm = {tuple(x): v for x, v in zip(r["pairs"].tolist(), r["ms"])}
                    dmax = max((abs(gm[x] - rm[x]) for x in common), default=0.0)
                    print(f"  pair {p}: S {S[p]} vs {r['S']}, symmetric difference {len(gp ^ rp)} (all borderline: {near}), max |dms| on common {dmax:.2e}", flush=True)
                    good = near and dmax < 5e-4
                ok &= good
            tag = f"lg P={P} Mmax={Mmax} Nmax={Nmax} m={ms_} n={ns_} fp16x2={int(h2)}"
        elif kind == 3:                                           # ---- sparse stereo matching on extracted features
            H, W = 8 * int(rng.integers(12, 40)) + int(rng.integers(0, 8)) * int(rng.random() < 0.4), 8 * int(rng.integers(16, 60)) + int(rng.integers(0, 8)) * int(rng.random() < 0.4)
            disp = int(rng.integers(0, 30))
            scene = synth.make_scene(rng, H, W + disp, margin=0)
            left = np.ascontiguousarray(np.clip(scene[:, :W] + rng.integers(0, 8, (H, W)), 0, 255).astype(np.uint8))
            right = np.ascontiguousarray(np.clip(scene[:, disp:disp + W] + rng.integers(0, 8, (H, W)), 0, 255).astype(np.uint8))
            c.set_weights(capi.KIND_SUPERPOINT, wsp)
            K = int(rng.choice([50, 200, 500]))
            n, kxy, score, desc = c.extract(np.stack([left, right]), kmax=K)
            kl, kr = kxy[0, :n[0]].astype(np.float32), kxy[1, :n[1]].astype(np.float32)
            mb = float(rng.uniform(0.05, 0.5)); mbf = mb * float(rng.uniform(200, 600))
            u, z = c.stereo_match(left, right, kl, kr, desc[0, :n[0]], desc[1, :n[1]], mb, mbf)
            ur, zr = O.stereo_match(left, right, kl, kr, desc[0, :n[0]], desc[1, :n[1]], mb, mbf)
            ok = bool(np.array_equal(u, ur) and np.array_equal(z, zr))
            tag = f"stereo H={H} W={W} disp={disp} n={n.tolist()}"
        elif kind == 4:                                           # ---- classic-search helpers
            Nq, Nf = int(rng.integers(1, 300)), int(rng.integers(1, 700))
            f = rng.standard_normal((Nf, 256)).astype(np.float32); f /= np.linalg.norm(f, axis=1, keepdims=True)
            q = f[rng.integers(0, Nf, Nq)] + 0.05 * rng.standard_normal((Nq, 256)).astype(np.float32)
            lens = rng.integers(0, 30, Nq); off = np.zeros(Nq + 1, np.int32); off[1:] = np.cumsum(lens)
            cand = rng.integers(0, Nf, off[-1]).astype(np.int32)
            skip = (rng.random(Nf) < 0.2).astype(np.uint8) if rng.random() < 0.7 else None
            got = c.search_candidates(q, f, off, cand, skip); ref = O.search_candidates(q, f, off, cand, skip)
            ok = all(np.array_equal(a, b) for a, b in zip(got, ref))
            plens = rng.integers(0, 40, int(rng.integers(1, 60))).astype(np.int32); poff = np.zeros(len(plens) + 1, np.int32); poff[1:] = np.cumsum(plens)
            dd = rng.standard_normal((max(int(poff[-1]), 1), 256)).astype(np.float32)
            got = c.distinctive_descriptors(dd, poff); ref = O.distinctive_descriptors(dd, poff)
            ok &= all(np.array_equal(a, b) for a, b in zip(got, ref))
            tag = f"search Nq={Nq} Nf={Nf} / distinctive Np={len(plens)}"
        else:                                                     # ---- stream mode vs extract + oracle matches
            big = rng.random() < 0.08
            H, W = (480, 640) if big else (8 * int(rng.integers(10, 31)) + int(rng.integers(0, 8)) * int(rng.random() < 0.4), 8 * int(rng.integers(10, 41)) + int(rng.integers(0, 8)) * int(rng.random() < 0.4))
            B, K = int(rng.integers(2, 8)), int(rng.choice([32, 48, 100, 128, 256, 300, 512, 1024] if big else [1, 5, 32, 33, 48, 100, 101, 128, 256, 300, 512]))
            c.set_weights(capi.KIND_SUPERPOINT, wsp)
            frames, _ = synth.make_frames(B, H, W, seed=int(rng.integers(1 << 30)))
            dimg = c.alloc(frames.nbytes).upload(frames)
            sizes = (B * 4, B * K * 8, B * K * 4, B * K * 1024, (B - 1) * 4, (B - 1) * K * 8, (B - 1) * K * 4)
            d = [c.alloc(s) for s in sizes]
            c._chk(capi.lib.rfe_extract_match_stream_dev(c.h, dimg.ptr, H, W, W, B, K, 0.0005, 0.1, *[x.ptr for x in d]))
            c.synchronize()
            n = d[0].download((B,), np.int32); kxy = d[1].download((B, K, 2), np.int32); desc = d[3].download((B, K, 256), np.float32)
            S = d[4].download((B - 1,), np.int32); pairs = d[5].download((B - 1, K, 2), np.int32)
            for x in d + [dimg]:
                x.free()
            n2, kxy2, _, desc2 = c.extract(frames, kmax=K)
            ok = bool(np.array_equal(n, n2) and np.array_equal(kxy, kxy2) and np.array_equal(desc, desc2))
            if not ok:
                print("  stream features differ from rfe_extract_u8", flush=True)
            for i in range(B - 1):
                if n[i] == 0 or n[i + 1] == 0:
                    ok &= bool(S[i] == 0)
                    continue
                k0n = O.normalize_keypoints(kxy[i, :n[i]].astype(np.float32), H, W); k1n = O.normalize_keypoints(kxy[i + 1, :n[i + 1]].astype(np.float32), H, W)
                r = O.lightglue(wlg, k0n, k1n, desc[i, :n[i]], desc[i + 1, :n[i + 1]], debug=True)
                good = bool(S[i] == r["S"] and np.array_equal(pairs[i, :S[i]], r["pairs"]))
                if not good:
                    S1, p1, m1 = c.match(k0n[None], k1n[None], desc[i, :n[i]][None], desc[i + 1, :n[i + 1]][None], [n[i]], [n[i + 1]])
                    gp = {tuple(x) for x in pairs[i, :S[i]].tolist()}; rp = {tuple(x) for x in r["pairs"].tolist()}
                    sc = r["scores"]
                    info = [(ij, float(np.exp(sc[ij])), top2_gap(sc[ij[0]]), top2_gap(sc[:, ij[1]])) for ij in gp ^ rp]
                    borderline_all = all(borderline(sc, ij[0], ij[1], max(n[i], n[i + 1])) for ij in gp ^ rp)   # may legitimately flip
                    print(f"  pair {i}: stream S {S[i]} oracle {r['S']} single-pair call {S1[0]} (== oracle: {np.array_equal(p1[0, :S1[0]], r['pairs'])}); "
                          f"borderline: {borderline_all}; differing (ij, exp(score), row gap, col gap in probability): {info}", flush=True)
                    good = borderline_all
                ok &= good
            tag = f"stream H={H} W={W} B={B} K={K} n={n.tolist()}"
        if not ok:
            fails += 1
            print("MISMATCH", tag, flush=True)
    print(f"fuzz: {it} cases in {time.time() - t0:.0f} s, {fails} mismatches")
    c.close()
    return fails


if __name__ == "__main__":
    a = sys.argv[1:]
    sys.exit(1 if main(float(a[0]) if a else 60.0, int(a[1]) if len(a) > 1 else 0) else 0)
