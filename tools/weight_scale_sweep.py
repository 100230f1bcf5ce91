#!/usr/bin/env python3
"""Where does the fp32 HIP path first leave its stated tolerances when the weights are not the tame seeded ones?  (VERDICT r03 item 7:
the cheapest stand-in for "real weights" available without the reference's blobs.)

Every Linear / conv FAMILY of the synthetic sets is scaled by 0.25x .. 8x in turn (final_proj up to similarity logits of several hundred),
and the same inputs are matched by the CPU oracle, by a float64 numpy evaluation of the graph, and by librover_fe.so -- one pair per call
(the latency tiling: gemm_lat / lg_attention_lat) and inside a 16-pair call (the throughput tiling: 128 x 256 GEMM tiles with the fused
LayerNorm + GELU, register-staged / LDS-DMA attention).  Reported per (family, scale): number of matches, largest similarity logit,
whether the match lists agree under tests/tolerances.py's borderline rule, max |match-score difference| HIP vs oracle / vs float64 and
oracle vs float64, how often the attention's deferred rescale and the short erf matter is visible in the deviations themselves.
SuperPoint rows: conv families scaled the same way; keypoints / scores / descriptors must stay BIT-exact (canonical arithmetic).

Test infrastructure (uses oracle/); GPU box:  python tools/weight_scale_sweep.py > profiles/rNN_weight_scale.md
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rover_slam_amd import weights as Wt, synth  # noqa: E402

LG_FAMILIES = {
    "attention q/k (self Wqkv rows 0..511, cross Wqk)": lambda n: n.endswith("self.Wqkv") or n.endswith("cross.Wqk"),
    "attention v / out (cross Wv, Wo of both blocks)": lambda n: n.endswith("cross.Wv") or n.endswith(".Wo"),
    "ffn.0 (W1, both blocks)": lambda n: n.endswith(".W1"),
    "ffn.3 (W2, both blocks)": lambda n: n.endswith(".W2"),
    "final_proj": lambda n: n == "final_proj.W",
    "posenc Wr": lambda n: n == "posenc.Wr",
}
SCALES = [0.25, 0.5, 2.0, 4.0, 8.0]


def scaled_lightglue(base, pred, s):
    blob = base.copy()
    for name, off, shape in Wt.lg_manifest()[0]:
        if not pred(name):
            continue
        cnt = int(np.prod(shape))
        if name.endswith("self.Wqkv"):
            cnt = 512 * 256                        # q and k rows only: v is the next family
        blob[off:off + cnt] *= np.float32(s)
    return blob


def case_inputs(K, seed, ragged):
    rng = np.random.default_rng(seed)
    m_ = int(rng.integers(K // 2, K)) if ragged else K
    d0 = rng.standard_normal((K, 256)).astype(np.float32)
    d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    perm = rng.permutation(K)
    d1 = d0[perm] + 0.01 * rng.standard_normal((K, 256)).astype(np.float32)
    d1 = (d1 / np.linalg.norm(d1, axis=1, keepdims=True)).astype(np.float32)
    k0 = rng.uniform(-0.9, 0.9, (K, 2)).astype(np.float32)
    k1 = (k0[perm] + 0.02 * rng.standard_normal((K, 2))).astype(np.float32)
    return np.ascontiguousarray(k0[:m_]), k1, np.ascontiguousarray(d0[:m_]), d1


def main():
    from rover_slam_amd import capi
    from oracle import oracle as O
    from lg_tolerance_study import lg_f64, score_dev
    from tolerances import lists_agree_borderline, LG_SCORE_TOL
    O.build()
    K, PB = 1024, 16
    ctx = capi.Context(0)
    base = Wt.make_lightglue(seed=11)
    inputs = [case_inputs(K, 900, False), case_inputs(K, 901, True)]
    t0 = time.time()
    rows = []
    todo = [("baseline (seeded weights)", None, 1.0)] + [(fam, pred, s) for fam, pred in LG_FAMILIES.items() for s in SCALES]
    for fam, pred, s in todo:
        blob = base if pred is None else scaled_lightglue(base, pred, s)
        ctx.set_weights(capi.KIND_LIGHTGLUE, blob)
        worst = {"S": 0, "logit": 0.0, "ok1": True, "okb": True, "dev1": 0.0, "devb": 0.0, "dev1_64": 0.0, "devo_64": 0.0, "one_sided": 0, "finite": True}
        for (k0, k1, d0, d1) in inputs:
            ref = O.lightglue(blob, k0, k1, d0, d1, debug=True)
            p64, m64, _ = lg_f64(blob, k0, k1, d0, d1)
            m, n = len(k0), len(k1)
            S, pairs, ms = ctx.match(k0[None], k1[None], d0[None], d1[None], [m], [n])
            ok1, dev1, only1 = lists_agree_borderline(pairs[0, :S[0]], ms[0, :S[0]], ref["pairs"], ref["ms"], ref["scores"], K)
            kb0 = np.zeros((PB, K, 2), np.float32); kb1 = np.zeros((PB, K, 2), np.float32)
            db0 = np.zeros((PB, K, 256), np.float32); db1 = np.zeros((PB, K, 256), np.float32)
            for i in range(PB):
                kb0[i, :m] = k0; kb1[i, :n] = k1; db0[i, :m] = d0; db1[i, :n] = d1
            Sb, pb, mb = ctx.match(kb0, kb1, db0, db1, [m] * PB, [n] * PB)
            okb, devb, onlyb = lists_agree_borderline(pb[PB - 1, :Sb[PB - 1]], mb[PB - 1, :Sb[PB - 1]], ref["pairs"], ref["ms"], ref["scores"], K)
            _, d164 = score_dev(pairs[0, :S[0]], ms[0, :S[0]], p64, m64)
            _, do64 = score_dev(ref["pairs"], ref["ms"], p64, m64)
            sim_peak = float(np.abs(ref["scores"][np.isfinite(ref["scores"])]).max()) if ref["scores"].size else 0.0
            worst["S"] = max(worst["S"], int(ref["S"])); worst["logit"] = max(worst["logit"], sim_peak)
            worst["ok1"] &= bool(ok1); worst["okb"] &= bool(okb)
            worst["dev1"] = max(worst["dev1"], dev1); worst["devb"] = max(worst["devb"], devb)
            worst["dev1_64"] = max(worst["dev1_64"], d164); worst["devo_64"] = max(worst["devo_64"], do64)
            worst["one_sided"] += only1 + onlyb
            worst["finite"] &= bool(np.isfinite(ms[0, :S[0]]).all() and np.isfinite(mb[PB - 1, :Sb[PB - 1]]).all())
        rows.append((fam, s, worst))
        print(f"# {fam} x{s}: {worst}", file=sys.stderr, flush=True)

    # SuperPoint: conv families scaled; bit-exactness of keypoints / scores / descriptors
    sp_rows = []
    frames, _ = synth.make_frames(1, 240, 320, seed=77)
    wsp0 = Wt.make_superpoint(seed=7)
    man = Wt.sp_manifest()[0]
    for fam, layers in (("encoder conv1a..conv4b", ("conv1", "conv2", "conv3", "conv4")), ("detector head convPa / convPb", ("convP",)),
                        ("descriptor head convDa / convDb", ("convD",))):
        for s in (0.25, 0.5, 2.0, 4.0):
            blob = wsp0.copy()
            for name, off, shape in man:
                if name.endswith(".weight") and name.startswith(layers):
                    blob[off:off + int(np.prod(shape))] *= np.float32(s)
            ctx.set_weights(capi.KIND_SUPERPOINT, blob)
            n, kxy, score, desc = ctx.extract(frames, kmax=1024)
            r = O.superpoint(blob, frames[0], kmax=1024)
            exact = bool(n[0] == r["n"] and np.array_equal(kxy[0], r["kxy"]) and np.array_equal(score[0], r["score"]) and np.array_equal(desc[0], r["desc"]))
            sp_rows.append((fam, s, int(r["n"]), exact, bool(np.isfinite(desc[0]).all())))
            print(f"# superpoint {fam} x{s}: n = {r['n']} exact = {exact}", file=sys.stderr, flush=True)
    ctx.close()

    yn = lambda b: "yes" if b else "**NO**"
    print("# Weight-scale sweep of the fp32 path (tools/weight_scale_sweep.py)\n")
    print(f"Seeded LightGlue weights (seed 11) with ONE family of Linear weights scaled at a time; two constructed 1024-keypoint cases per row "
          f"(permuted noisy copies with hundreds of true matches, one of them ragged); `1 pair` = one pair per call (latency tiling), `16 pairs` = the "
          f"same pair inside a 16-pair call (throughput tiling).  `lists agree` is tests/tolerances.py's rule (a one-sided match must be borderline), "
          f"deviations are max |match-score difference| over common matches; the stated tolerance is {LG_SCORE_TOL:g}.  `peak |log-score|` = largest "
          f"finite magnitude in the oracle's log-assignment matrix.  {time.time() - t0:.0f} s.\n")
    print("| family | scale | matches | peak \\|log-score\\| | 1 pair: lists agree | HIP vs oracle | 16 pairs: lists agree | HIP vs oracle | one-sided (borderline) | HIP vs f64 (1 pair) | oracle vs f64 | finite |")
    print("|---|---:|---:|---:|:-:|---:|:-:|---:|---:|---:|---:|:-:|")
    for fam, s, w in rows:
        flag = "" if max(w["dev1"], w["devb"]) < LG_SCORE_TOL else " ⚠"
        print(f"| {fam} | {s:g} | {w['S']} | {w['logit']:.0f} | {yn(w['ok1'])} | {w['dev1']:.2e}{flag} | {yn(w['okb'])} | {w['devb']:.2e} | {w['one_sided']} "
              f"| {w['dev1_64']:.2e} | {w['devo_64']:.2e} | {yn(w['finite'])} |")
    print("\n## SuperPoint (240 x 320 frame, Kmax 1024): conv weights of a family scaled, outputs against the oracle\n")
    print("| family | scale | keypoints | keypoints / scores / descriptors bit-exact | finite |")
    print("|---|---:|---:|:-:|:-:|")
    for fam, s, n, exact, fin in sp_rows:
        print(f"| {fam} | {s:g} | {n} | {yn(exact)} | {yn(fin)} |")


if __name__ == "__main__":
    main()
