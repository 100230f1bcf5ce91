#!/usr/bin/env python3
"""Parity of the oracle (and, with --gpu, of librover_fe.so) against the TRUE reference arithmetic:
ONNXRuntime executing the reference's own onnxmodel/superpoint.onnx and onnxmodel/lightglue_sim.onnx
(reference call sites: src/Extractors/superpoint_onnx.cc:133-136, src/Matchers/lightglue_onnx.cpp:210-214).

SURVEY.md 8(f) N1 / 8(c): neither onnxruntime nor the two blobs exist in the build image, so this tool
cannot run there -- it is the harness that turns "parity unpinned" into a checked claim the moment a user
has both.  Everything reference-specific is probed at run time; nothing is imported at module load.

    python tools/ort_parity.py --superpoint onnxmodel/superpoint.onnx --lightglue onnxmodel/lightglue_sim.onnx [--gpu]

Per frame it reports: keypoint set equality, max |score| and max descriptor L2 deviation (bar: keypoints and
scores identical after NMS, descriptors <= 1e-4); per pair: match-list equality and max |mscore| deviation.
Exit code 0 = within the bars, 1 = deviation, 2 = prerequisites missing.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--superpoint", required=True)
    ap.add_argument("--lightglue")
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--gpu", action="store_true", help="also run librover_fe.so on cuda:0")
    ap.add_argument("--desc-tol", type=float, default=1e-4)
    ap.add_argument("--assume-sp", action="append", metavar="KEY=VALUE", help="state a SuperPoint hyper-parameter the graph does not reveal")
    ap.add_argument("--assume-lg", action="append", metavar="KEY=VALUE", help="the same for LightGlue")
    a = ap.parse_args(argv)
    try:
        import onnxruntime as ort
    except ImportError:
        print("ort_parity: onnxruntime is not installed; parity against the reference stays unpinned", file=sys.stderr)
        return 2
    for p in (a.superpoint, a.lightglue):
        if p and not os.path.exists(p):
            print(f"ort_parity: {p} not found", file=sys.stderr)
            return 2
    from rover_slam_amd import onnx_weights, synth
    from oracle import oracle
    oracle.build()
    frames, _ = synth.make_frames(a.frames, 480, 640)
    # the graph's baked-in hyper-parameters first: a deviation below must be arithmetic, not a silently different K / radius / threshold
    read, problems = onnx_weights.read_superpoint_hparams(a.superpoint)
    print(f"{a.superpoint}: hyper-parameters read from the graph: {read}" + (f"; unresolved: {problems}" if problems else ""))
    try:
        wsp, hp = onnx_weights.convert(a.superpoint, 1, onnx_weights._parse_assume(a.assume_sp))
    except ValueError as e:
        print(f"ort_parity: {e}\n  (state what the graph does not reveal with --assume-sp KEY=VALUE)", file=sys.stderr)
        return 2
    sp = ort.InferenceSession(a.superpoint, providers=["CPUExecutionProvider"])
    ctx = None
    if a.gpu:
        from rover_slam_amd import capi
        ctx = capi.Context(0)
        ctx.set_weights(capi.KIND_SUPERPOINT, wsp)
        ctx.set_hparams(sp_max_keypoints=hp["max_keypoints"], sp_detection_threshold=hp["detection_threshold"], sp_nms_radius=hp["nms_radius"],
                        sp_remove_borders=hp["remove_borders"], sp_topk_always=hp["topk_always"])
    bad = False
    feats = []
    for i, img in enumerate(frames):
        x = (img.astype(np.float32) / 255.0)[None, None]                       # NormalizeImage, transform.cpp:3-17
        k_ref, s_ref, d_ref = sp.run(["keypoints", "scores", "descriptors"], {"image": x})
        k_ref, s_ref, d_ref = k_ref[0].astype(np.int64), s_ref[0], d_ref[0]
        K = k_ref.shape[0]
        o = oracle.superpoint(wsp, img, kmax=hp["max_keypoints"], thr=hp["detection_threshold"], nms_radius=hp["nms_radius"],
                              border=hp["remove_borders"], topk_always=bool(hp["topk_always"]))
        cands = [("oracle", o["n"], o["kxy"], o["score"], o["desc"])]
        if ctx is not None:
            n, kxy, sc, de = ctx.extract(img[None], kmax=hp["max_keypoints"], thr=hp["detection_threshold"])
            cands.append(("hip", int(n[0]), kxy[0], sc[0], de[0]))
        for name, n, kxy, sc, de in cands:
            ref = {(int(x_), int(y_)): j for j, (x_, y_) in enumerate(k_ref)}
            got = {(int(x_), int(y_)): j for j, (x_, y_) in enumerate(kxy[:n])}
            same = set(ref) == set(got)
            common = sorted(set(ref) & set(got))
            ds = max((abs(float(s_ref[ref[c]]) - float(sc[got[c]])) for c in common), default=0.0)
            dd = max((float(np.linalg.norm(d_ref[ref[c]] - de[got[c]])) for c in common), default=0.0)
            order = same and all(ref[c] == got[c] for c in common)
            print(f"frame {i} {name}: K_ref={K} K={n} same_set={same} same_order={order} max|dscore|={ds:.3g} max desc L2={dd:.3g}")
            bad |= (not same) or dd > a.desc_tol
        feats.append((k_ref, d_ref))
    if a.lightglue:
        read, problems = onnx_weights.read_lightglue_hparams(a.lightglue)
        print(f"{a.lightglue}: hyper-parameters read from the graph: {read}" + (f"; unresolved: {problems}" if problems else ""))
        try:
            wlg, hpl = onnx_weights.convert(a.lightglue, 2, onnx_weights._parse_assume(a.assume_lg))
        except ValueError as e:
            print(f"ort_parity: {e}\n  (state what the graph does not reveal with --assume-lg KEY=VALUE)", file=sys.stderr)
            return 2
        lg = ort.InferenceSession(a.lightglue, providers=["CPUExecutionProvider"])
        if ctx is not None:
            ctx.set_weights(capi.KIND_LIGHTGLUE, wlg)
        for i in range(len(feats) - 1):
            (k0, d0), (k1, d1) = feats[i], feats[i + 1]
            k0n = oracle.normalize_keypoints(k0.astype(np.float32), 480, 640)   # NormalizeKeypoints, transform.cpp:19-32
            k1n = oracle.normalize_keypoints(k1.astype(np.float32), 480, 640)
            m_ref, ms_ref = lg.run(["matches0", "mscores0"], {"kpts0": k0n[None], "kpts1": k1n[None], "desc0": d0[None], "desc1": d1[None]})
            o = oracle.lightglue(wlg, k0n, k1n, d0, d1, filter_thr=hpl["filter_threshold"])
            cands = [("oracle", o["pairs"], o["ms"])]
            if ctx is not None:
                S, pairs, ms = ctx.match(k0n[None], k1n[None], d0[None], d1[None], [len(k0n)], [len(k1n)], filter_thr=hpl["filter_threshold"])
                cands.append(("hip", pairs[0, :S[0]], ms[0, :S[0]]))
            for name, pairs, ms in cands:
                same = pairs.shape == m_ref.shape and np.array_equal(pairs, m_ref)
                dm = float(np.abs(ms - ms_ref).max()) if same and len(ms) else float("nan")
                print(f"pair {i} {name}: S_ref={len(m_ref)} S={len(pairs)} same_matches={same} max|dmscore|={dm:.3g}")
                bad |= not same
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
