#!/usr/bin/env python3
"""How far do LightGlue match scores move between arithmetically different but mathematically identical evaluations?

For 40 (weight seed, input) cases at the bench shape (Kmax = 1024; even cases: two consecutive synthetic 640x480 frames
through SuperPoint, odd cases: constructed sets with hundreds of true matches, every other one ragged) the same pair is matched by
  * the CPU oracle (fp32, the restatement the parity tests compare against),
  * a float64 numpy evaluation of the same graph (this file; "exact" for the purpose of fp32 rounding),
  * the HIP path with RFE_OPT_LG_FOLD_WO = 0 (graph node for node) and = 1 (Wo folded into ffn.0),
and the maximum absolute match-score deviation between them is tabulated, together with whether the match lists are
identical.  Test infrastructure (uses oracle/); run on the GPU box:  python tools/lg_tolerance_study.py [--cases 20] [--batched]
Writes a markdown table to stdout (committed as profiles/rNN_lg_tolerance.md).
--batched (round 3): every case is ALSO matched inside a 16-pair call (the cases that share a LightGlue weight seed, cycled up to 16
pairs = 32 768 token rows), i.e. through the THROUGHPUT tiling the benchmark times -- 128x256 k-permuted GEMM tiles, LayerNorm + GELU
fused across ffn.0 / ffn.3, the register-staged self attention and the LDS-DMA cross attention with one workgroup per query block --
where the plain study (one pair per call) takes the 64-row latency tiles and the split-key attention.
"""
import argparse
import os
import sys
import time

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rover_slam_amd import weights as Wt, synth  # noqa: E402


def lg_f64(blob, k0n, k1n, d0, d1, thr=0.1):
    """LightGlue in float64 numpy.  Returns pairs [S,2], ms [S], scores [M,N]."""
    man, _ = Wt.lg_manifest()
    t = {name: blob[off:off + int(np.prod(shape))].reshape(shape).astype(np.float64) for name, off, shape in man}

    def posenc(kn):
        th = kn.astype(np.float64) @ t["posenc.Wr"].T            # [n,32]
        return np.cos(th), np.sin(th)

    def rotary(x, cs, sn):                                       # x [n,256] head-major 4x64, pairs (2f, 2f+1)
        n = x.shape[0]
        v = x.reshape(n, 4, 32, 2)
        a, b = v[..., 0], v[..., 1]
        c, s = cs[:, None, :], sn[:, None, :]
        return np.stack([a * c - b * s, b * c + a * s], -1).reshape(n, 256)

    def attention(q, k, v):
        nq, nk = q.shape[0], k.shape[0]
        qh, kh, vh = (z.reshape(-1, 4, 64).transpose(1, 0, 2) for z in (q, k, v))
        s = np.einsum("hid,hjd->hij", qh, kh) * 0.125
        s -= s.max(-1, keepdims=True)
        p = np.exp(s)
        p /= p.sum(-1, keepdims=True)
        return np.einsum("hij,hjd->hid", p, vh).transpose(1, 0, 2).reshape(nq, 256)

    def ffn(x, msg, p):
        h = np.concatenate([x, msg], 1) @ t[p + "W1"].T + t[p + "b1"]
        mu = h.mean(1, keepdims=True)
        var = ((h - mu) ** 2).mean(1, keepdims=True)
        h = (h - mu) / np.sqrt(var + 1e-5) * t[p + "ln_g"] + t[p + "ln_b"]
        h = 0.5 * h * (1.0 + erf(h * 0.70710678118654752))
        return x + h @ t[p + "W2"].T + t[p + "b2"]

    x0, x1 = d0.astype(np.float64), d1.astype(np.float64)
    (c0, s0), (c1, s1) = posenc(k0n), posenc(k1n)
    for l in range(9):
        p = f"layers.{l}.self."
        for side in (0, 1):
            x, cs, sn = (x0, c0, s0) if side == 0 else (x1, c1, s1)
            qkv = x @ t[p + "Wqkv"].T + t[p + "bqkv"]
            q, k, v = rotary(qkv[:, :256], cs, sn), rotary(qkv[:, 256:512], cs, sn), qkv[:, 512:]
            msg = attention(q, k, v) @ t[p + "Wo"].T + t[p + "bo"]
            x = ffn(x, msg, p)
            if side == 0:
                x0 = x
            else:
                x1 = x
        p = f"layers.{l}.cross."
        qk0, qk1 = x0 @ t[p + "Wqk"].T + t[p + "bqk"], x1 @ t[p + "Wqk"].T + t[p + "bqk"]
        v0, v1 = x0 @ t[p + "Wv"].T + t[p + "bv"], x1 @ t[p + "Wv"].T + t[p + "bv"]
        m0 = attention(qk0, qk1, v1) @ t[p + "Wo"].T + t[p + "bo"]
        m1 = attention(qk1, qk0, v0) @ t[p + "Wo"].T + t[p + "bo"]
        x0, x1 = ffn(x0, m0, p), ffn(x1, m1, p)
    md0 = (x0 @ t["final_proj.W"].T + t["final_proj.b"]) * 0.25
    md1 = (x1 @ t["final_proj.W"].T + t["final_proj.b"]) * 0.25
    sim = md0 @ md1.T
    z0, z1 = x0 @ t["matchability.w"] + t["matchability.b"], x1 @ t["matchability.w"] + t["matchability.b"]
    logsig = lambda z: -np.logaddexp(0.0, -z)
    lse = lambda a, ax: a.max(ax, keepdims=True) + np.log(np.exp(a - a.max(ax, keepdims=True)).sum(ax, keepdims=True))
    sc = (sim - lse(sim, 1)) + (sim - lse(sim, 0)) + logsig(z0)[:, None] + logsig(z1)[None, :]
    a0, a1 = sc.argmax(1), sc.argmax(0)
    i = np.arange(sc.shape[0])
    e = np.exp(sc[i, a0])
    keep = (a1[a0] == i) & (e > thr)
    return np.stack([i[keep], a0[keep]], 1).astype(np.int32), e[keep], sc


def score_dev(pa, ma, pb, mb):
    """(lists identical?, max |score diff| over the matches both lists share)."""
    da = {(int(i), int(j)): float(s) for (i, j), s in zip(pa, ma)}
    db = {(int(i), int(j)): float(s) for (i, j), s in zip(pb, mb)}
    common = da.keys() & db.keys()
    dev = max((abs(da[k] - db[k]) for k in common), default=0.0)
    return len(da) == len(db) == len(common), dev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--kmax", type=int, default=1024)
    ap.add_argument("--batched", action="store_true", help="also match every case inside a 16-pair call (throughput tiling)")
    args = ap.parse_args()
    from rover_slam_amd import capi
    from oracle import oracle as O
    O.build()
    H, W, K = 480, 640, args.kmax
    ctx = capi.Context(0)
    wsp = Wt.make_superpoint(seed=7)
    ctx.set_weights(capi.KIND_SUPERPOINT, wsp)
    rows = []
    t_start = time.time()
    for case in range(args.cases):
        lg_seed, fr_seed = 11 + case % 4, 500 + case
        wlg = Wt.make_lightglue(seed=lg_seed)
        ctx.set_weights(capi.KIND_LIGHTGLUE, wlg)
        if case % 2 == 0:      # two consecutive synthetic frames through SuperPoint: real features, few matches (random weights)
            frames, _ = synth.make_frames(2, H, W, seed=fr_seed)
            n, kxy, score, desc = ctx.extract(frames, kmax=K)
            k0 = O.normalize_keypoints(kxy[0, :n[0]].astype(np.float32), H, W)
            k1 = O.normalize_keypoints(kxy[1, :n[1]].astype(np.float32), H, W)
            d0, d1 = desc[0, :n[0]], desc[1, :n[1]]
        else:                  # constructed: set 1 = permuted noisy copy of a random unit-vector set 0 -> hundreds of matches
            rng = np.random.default_rng(fr_seed)
            m_ = K if case % 4 == 1 else int(rng.integers(K // 2, K))
            d0 = rng.standard_normal((K, 256)).astype(np.float32)
            d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
            perm = rng.permutation(K)
            d1 = d0[perm] + 0.01 * rng.standard_normal((K, 256)).astype(np.float32)
            d1 = (d1 / np.linalg.norm(d1, axis=1, keepdims=True)).astype(np.float32)
            k0 = rng.uniform(-0.9, 0.9, (K, 2)).astype(np.float32)
            k1 = (k0[perm] + 0.02 * rng.standard_normal((K, 2))).astype(np.float32)
            k0, d0 = np.ascontiguousarray(k0[:m_]), np.ascontiguousarray(d0[:m_])
            n = np.array([m_, K], np.int32)
        ref = O.lightglue(wlg, k0, k1, d0, d1)
        p64, m64, _ = lg_f64(wlg, k0, k1, d0, d1)
        got = {}
        for fold in (0, 1):
            ctx.set_option(capi.OPT_LG_FOLD_WO, fold)
            S, pairs, ms = ctx.match(k0[None], k1[None], d0[None], d1[None], [n[0]], [n[1]])
            got[fold] = (pairs[0, :S[0]], ms[0, :S[0]])
        r = {"case": case, "lg_seed": lg_seed, "frames_seed": fr_seed, "n0": int(n[0]), "n1": int(n[1]), "S": ref["S"]}
        for fold in (0, 1):
            r[f"same{fold}"], r[f"dev{fold}"] = score_dev(got[fold][0], got[fold][1], ref["pairs"], ref["ms"])
            r[f"same{fold}_64"], r[f"dev{fold}_64"] = score_dev(got[fold][0], got[fold][1], p64, m64)
        # one pair per call with RFE_OPT_LG_FP16X2 = 1: the split forms of the latency kernels (gemm_lat.hip / lg_attention_lat.hip, H2), default fold
        ctx.set_option(capi.OPT_LG_FP16X2, 1)
        try:
            S, pairs, ms = ctx.match(k0[None], k1[None], d0[None], d1[None], [n[0]], [n[1]])
        finally:
            ctx.set_option(capi.OPT_LG_FP16X2, 0)
        r["sameh"], r["devh"] = score_dev(pairs[0, :S[0]], ms[0, :S[0]], ref["pairs"], ref["ms"])
        r["sameh_64"], r["devh_64"] = score_dev(pairs[0, :S[0]], ms[0, :S[0]], p64, m64)
        r["same_o64"], r["dev_o64"] = score_dev(ref["pairs"], ref["ms"], p64, m64)
        r["_in"] = (k0, k1, d0, d1, ref, p64, m64)
        rows.append(r)
        print(f"# case {case}: {r}", file=sys.stderr, flush=True)
    if args.batched:
        PB = 16
        for seed in sorted({r["lg_seed"] for r in rows}):
            grp = [r for r in rows if r["lg_seed"] == seed]
            ctx.set_weights(capi.KIND_LIGHTGLUE, Wt.make_lightglue(seed=seed))
            for g0 in range(0, len(grp), PB):
                part = grp[g0:g0 + PB]
                cyc = [part[i % len(part)] for i in range(PB)]           # cycled up to 16 pairs: >= 32768 token rows
                kb0 = np.zeros((PB, K, 2), np.float32); kb1 = np.zeros((PB, K, 2), np.float32)
                db0 = np.zeros((PB, K, 256), np.float32); db1 = np.zeros((PB, K, 256), np.float32)
                for i, r in enumerate(cyc):
                    k0, k1, d0, d1 = r["_in"][:4]
                    kb0[i, :r["n0"]] = k0; kb1[i, :r["n1"]] = k1; db0[i, :r["n0"]] = d0; db1[i, :r["n1"]] = d1
                for fold in (0, 1):
                    ctx.set_option(capi.OPT_LG_FOLD_WO, fold)
                    S, pairs, ms = ctx.match(kb0, kb1, db0, db1, [r["n0"] for r in cyc], [r["n1"] for r in cyc])
                    for i, r in enumerate(part):
                        ref, p64, m64 = r["_in"][4:]
                        r[f"bsame{fold}"], r[f"bdev{fold}"] = score_dev(pairs[i, :S[i]], ms[i, :S[i]], ref["pairs"], ref["ms"])
                        r[f"bsame{fold}_64"], r[f"bdev{fold}_64"] = score_dev(pairs[i, :S[i]], ms[i, :S[i]], p64, m64)
                    # RFE_OPT_LG_FP16X2: the Linears of the same call as a split GEMM on the f16 matrix pipe (gemm_h2.hip)
                    ctx.set_option(capi.OPT_LG_FP16X2, 1)
                    S, pairs, ms = ctx.match(kb0, kb1, db0, db1, [r["n0"] for r in cyc], [r["n1"] for r in cyc])
                    ctx.set_option(capi.OPT_LG_FP16X2, 0)
                    for i, r in enumerate(part):
                        ref, p64, m64 = r["_in"][4:]
                        r[f"hsame{fold}"], r[f"hdev{fold}"] = score_dev(pairs[i, :S[i]], ms[i, :S[i]], ref["pairs"], ref["ms"])
                        r[f"hsame{fold}_64"], r[f"hdev{fold}_64"] = score_dev(pairs[i, :S[i]], ms[i, :S[i]], p64, m64)
            print(f"# batched seed {seed} done", file=sys.stderr, flush=True)
    ctx.close()
    mx = lambda k: max(r[k] for r in rows)
    al = lambda k: all(r[k] for r in rows)
    print("# LightGlue match-score tolerance study (tools/lg_tolerance_study.py)\n")
    print(f"{len(rows)} pairs (even cases: consecutive synthetic 640x480 frames through SuperPoint; odd cases: constructed permuted "
          f"noisy copies, every other one ragged), Kmax = {K}, LightGlue weight seeds 11..14, SuperPoint seed 7; "
          f"{time.time() - t_start:.0f} s.  Match scores are probabilities in (0.1, 1].  `oracle` = oracle/rfe_oracle.c (fp32), "
          "`f64` = the same graph in float64 numpy, `gpu0` / `gpu1` = HIP path with RFE_OPT_LG_FOLD_WO = 0 / 1.  "
          "A deviation is the max |score difference| over the matches both lists contain; `lists` = match lists identical.\n")
    print("| case | lg seed | n0 | n1 | S | gpu0 vs oracle | lists | gpu1 vs oracle | lists | oracle vs f64 | lists | gpu0 vs f64 | gpu1 vs f64 |")
    print("|---:|---:|---:|---:|---:|---:|:-:|---:|:-:|---:|:-:|---:|---:|")
    yn = lambda b: "yes" if b else "NO"
    for r in rows:
        print(f"| {r['case']} | {r['lg_seed']} | {r['n0']} | {r['n1']} | {r['S']} | {r['dev0']:.2e} | {yn(r['same0'])} | {r['dev1']:.2e} | {yn(r['same1'])} "
              f"| {r['dev_o64']:.2e} | {yn(r['same_o64'])} | {r['dev0_64']:.2e} | {r['dev1_64']:.2e} |")
    print(f"| **max** | | | | | **{mx('dev0'):.2e}** | {yn(al('same0'))} | **{mx('dev1'):.2e}** | {yn(al('same1'))} | **{mx('dev_o64'):.2e}** "
          f"| {yn(al('same_o64'))} | **{mx('dev0_64'):.2e}** | **{mx('dev1_64'):.2e}** |")
    print("\n## One pair per call with RFE_OPT_LG_FP16X2 = 1 (split forms of the latency kernels: Linears AND attention; RFE_OPT_LG_FOLD_WO at its default 1)\n")
    print("| case | S | fp16x2 vs oracle | lists | fp16x2 vs f64 | lists |")
    print("|---:|---:|---:|:-:|---:|:-:|")
    for r in rows:
        print(f"| {r['case']} | {r['S']} | {r['devh']:.2e} | {yn(r['sameh'])} | {r['devh_64']:.2e} | {yn(r['sameh_64'])} |")
    print(f"| **max** | | **{mx('devh'):.2e}** | {yn(al('sameh'))} | **{mx('devh_64'):.2e}** | {yn(al('sameh_64'))} |")
    if args.batched:
        print("\n## The same cases inside 16-pair calls (throughput tiling: >= 32 768 token rows per call)\n")
        print("| case | S | gpu0 batched vs oracle | lists | gpu1 batched vs oracle | lists | gpu0 batched vs f64 | lists | gpu1 batched vs f64 | lists |")
        print("|---:|---:|---:|:-:|---:|:-:|---:|:-:|---:|:-:|")
        for r in rows:
            print(f"| {r['case']} | {r['S']} | {r['bdev0']:.2e} | {yn(r['bsame0'])} | {r['bdev1']:.2e} | {yn(r['bsame1'])} | {r['bdev0_64']:.2e} | {yn(r['bsame0_64'])} "
                  f"| {r['bdev1_64']:.2e} | {yn(r['bsame1_64'])} |")
        print(f"| **max** | | **{mx('bdev0'):.2e}** | {yn(al('bsame0'))} | **{mx('bdev1'):.2e}** | {yn(al('bsame1'))} | **{mx('bdev0_64'):.2e}** | {yn(al('bsame0_64'))} "
              f"| **{mx('bdev1_64'):.2e}** | {yn(al('bsame1_64'))} |")
        print("\n## ... with RFE_OPT_LG_FP16X2 = 1 (16-pair calls: Linears and attention as fp16 hi + lo split products on the f16 matrix pipe, gemm_h2.hip / lg_attention_h2.hip; default off)\n")
        print("| case | S | gpu0 fp16x2 vs oracle | lists | gpu1 fp16x2 vs oracle | lists | gpu0 fp16x2 vs f64 | lists | gpu1 fp16x2 vs f64 | lists |")
        print("|---:|---:|---:|:-:|---:|:-:|---:|:-:|---:|:-:|")
        for r in rows:
            print(f"| {r['case']} | {r['S']} | {r['hdev0']:.2e} | {yn(r['hsame0'])} | {r['hdev1']:.2e} | {yn(r['hsame1'])} | {r['hdev0_64']:.2e} | {yn(r['hsame0_64'])} "
                  f"| {r['hdev1_64']:.2e} | {yn(r['hsame1_64'])} |")
        print(f"| **max** | | **{mx('hdev0'):.2e}** | {yn(al('hsame0'))} | **{mx('hdev1'):.2e}** | {yn(al('hsame1'))} | **{mx('hdev0_64'):.2e}** | {yn(al('hsame0_64'))} "
              f"| **{mx('hdev1_64'):.2e}** | {yn(al('hsame1_64'))} |")


if __name__ == "__main__":
    main()
