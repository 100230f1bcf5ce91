#!/usr/bin/env python3
"""Parity of the oracle (and, with --gpu, of librover_fe.so) against an EXECUTION OF THE GRAPH FILES the reference runs:
onnxmodel/superpoint.onnx and onnxmodel/lightglue_sim.onnx through `Session::Run`
(reference call sites: src/Extractors/superpoint_onnx.cc:133-136, src/Matchers/lightglue_onnx.cpp:210-214).

    python tools/ort_parity.py --superpoint onnxmodel/superpoint.onnx --lightglue onnxmodel/lightglue_sim.onnx [--gpu] [--backend ort|mini]

--backend ort  (default): onnxruntime's CPUExecutionProvider -- the TRUE reference arithmetic.  Neither onnxruntime nor the two blobs
               exist in the build image (.MISSING_LARGE_BLOBS:4-5); exit code 2 says so.
--backend replay --replay FILE.npz : the graph outputs a previous `--backend mini|ort --save FILE.npz` run wrote, replayed call by call (the feeds
               must be the saved ones) -- how the committed fixtures tests/golden/onnx_s*.npz (tools/gen_onnx_golden.py) reach the GPU box,
               where `-m gpu` tests run this harness with --gpu.
--backend mini : tools/mini_onnx.py, a numpy / torch-CPU interpreter of the same graph file (build container only).  With graphs written
               by torch's ONNX exporter from the published modules (tools/onnx_export.py) this executes EVERY line of this harness and
               pins the oracle against graph execution -- keypoint order, int64 layout, the (y, x) -> (x, y) flip, TopK ties, border,
               the in-graph match filter -- which the module-level tests cannot see.  tests/test_ort_parity.py runs it.

The weights and hyper-parameters the oracle / the HIP library use come out of the SAME file through rover_slam_amd.onnx_weights
(.onnx -> canonical blob + RFEW v2 hyper-parameters), so a deviation is arithmetic or graph semantics, never a different K / radius.

Per frame: keypoint SET equality (required), order (required up to swaps of keypoints whose reference scores differ by <= --score-tol),
max |score| deviation (<= --score-tol) and max descriptor L2 deviation (<= --desc-tol, north_star's 1e-4).  Per pair: match-list equality
(a match only one side reports must sit within --mscore-tol of the filter threshold) and max |mscore| deviation (<= --mscore-tol).
Exit code 0 = within the bars, 1 = deviation, 2 = prerequisites missing.  --save FILE writes every graph output (the fixture format of
tools/gen_onnx_golden.py).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


class _Arg:
    def __init__(self, name):
        self.name = name


class _ReplaySession:
    """Saved graph outputs behind the InferenceSession surface: run() hands back what the recorded execution returned for the SAME feed."""

    def __init__(self, z, kind):
        self.z, self.kind, self.calls = z, kind, 0

    def get_inputs(self):
        return [_Arg(n) for n in self.z[self.kind + "_inputs"].tolist()]

    def get_outputs(self):
        return [_Arg(n) for n in self.z[self.kind + "_outputs"].tolist()]

    def run(self, names, feeds):
        i, self.calls = self.calls, self.calls + 1
        if self.kind == "sp":
            want = (self.z["frames"][i].astype(np.float32) / 255.0)[None, None]
            if not np.array_equal(feeds["image"], want):
                raise ValueError(f"replay: frame {i} is not the recorded feed")
            return [self.z[f"sp{i}_{n}"][None] for n in names]
        want = {"kpts0": self.z[f"lg{i}_kpts0"], "kpts1": self.z[f"lg{i}_kpts1"],             # pair i = frames i, i + 1: the descriptors are the
                "desc0": self.z[f"sp{i}_descriptors"][None], "desc1": self.z[f"sp{i + 1}_descriptors"][None]}   # extractor graph's own outputs
        for k in want:
            if not np.array_equal(feeds[k], want[k]):
                raise ValueError(f"replay: pair {i} input {k} is not the recorded feed")
        return [self.z[f"lg{i}_{n}"] for n in names]


def _session_factory(backend, replay=None):
    """-> (make_session(path), description) or (None, reason)"""
    if backend == "replay":
        if not replay or not os.path.exists(replay):
            return None, f"--backend replay needs --replay FILE.npz ({replay!r} not found)"
        z = np.load(replay)
        return (lambda p: _ReplaySession(z, "sp" if "superpoint" in os.path.basename(p) else "lg")), \
            f"replay of {replay} (recorded from: {str(z['recorded_from'])})"
    if backend == "mini":
        import mini_onnx
        return (lambda p: mini_onnx.InferenceSession(p)), "tools/mini_onnx.py (numpy / torch-CPU graph interpreter; NOT the reference runtime)"
    try:
        import onnxruntime as ort
    except ImportError:
        return None, "onnxruntime is not installed; parity against the reference runtime stays unpinned (use --backend mini for graph-execution parity)"
    return (lambda p: ort.InferenceSession(p, providers=["CPUExecutionProvider"])), f"onnxruntime {ort.__version__} CPUExecutionProvider"


def weights_sha256(path):
    """sha256 over every float initializer of a graph file (name-sorted): identifies the weights a recording was made with"""
    import hashlib
    from rover_slam_amd import onnx_weights
    inits, _ = onnx_weights.read_model(path)
    h = hashlib.sha256()
    for k in sorted(inits):
        if inits[k].dtype == np.float32:
            h.update(np.ascontiguousarray(inits[k]).tobytes())
    return h.hexdigest()


def compare_keypoints(k_ref, s_ref, d_ref, n, kxy, sc, de, score_tol):
    """-> dict(same_set, order_ok, same_order, dscore, ddesc) for one frame: reference (graph) outputs vs a candidate's first n rows"""
    ref = {(int(x_), int(y_)): j for j, (x_, y_) in enumerate(k_ref)}
    got = {(int(x_), int(y_)): j for j, (x_, y_) in enumerate(kxy[:n])}
    same = set(ref) == set(got) and len(ref) == len(k_ref) and len(got) == n
    common = sorted(set(ref) & set(got))
    ds = max((abs(float(s_ref[ref[c]]) - float(sc[got[c]])) for c in common), default=0.0)
    dd = max((float(np.linalg.norm(d_ref[ref[c]] - de[got[c]])) for c in common), default=0.0)
    same_order = same and all(ref[c] == got[c] for c in common)
    # the candidate's order, read through the REFERENCE scores, may only deviate where those scores are within the tolerance of each other
    order_ok = same
    if same and not same_order:
        seq = np.array([float(s_ref[ref[(int(x_), int(y_))]]) for x_, y_ in kxy[:n]])
        srt = np.array([float(v) for v in s_ref])
        order_ok = bool(np.all(np.abs(seq - srt) <= score_tol))      # position by position the reference score is (nearly) the same
    return dict(same_set=bool(same), order_ok=bool(order_ok), same_order=bool(same_order), dscore=ds, ddesc=dd)


def compare_matches(m_ref, ms_ref, pairs, ms, thr, tol):
    da = {(int(i), int(j)): float(s) for (i, j), s in zip(m_ref, ms_ref)}
    db = {(int(i), int(j)): float(s) for (i, j), s in zip(pairs, ms)}
    only = [da.get(k, db.get(k)) for k in da.keys() ^ db.keys()]
    lists_ok = all(abs(s - thr) <= tol for s in only)
    dev = max((abs(da[k] - db[k]) for k in da.keys() & db.keys()), default=0.0)
    identical = len(m_ref) == len(pairs) and np.array_equal(np.asarray(pairs, np.int64).reshape(-1, 2), np.asarray(m_ref, np.int64).reshape(-1, 2))
    return dict(identical=bool(identical), lists_ok=bool(lists_ok), one_sided=len(only), dev=dev)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--superpoint", required=True)
    ap.add_argument("--lightglue")
    ap.add_argument("--backend", default="ort", choices=["ort", "mini", "replay"])
    ap.add_argument("--replay", metavar="FILE.npz")
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--frame-seed", type=int, default=20240314)
    ap.add_argument("--shift-step", type=int, default=1, help="8: consecutive frames shifted by multiples of the 8-px cell (many matches with seeded weights)")
    ap.add_argument("--gpu", action="store_true", help="also run librover_fe.so on cuda:0 (weights and hyper-parameters through an RFEW v2 file)")
    ap.add_argument("--gpu-onnx", action="store_true", help="with --gpu: hand the .onnx files themselves to rfe_load_weights (the library's C++ reader, "
                                                            "rover-slam_amd/csrc/onnx_load.hip) instead of going through an RFEW v2 container")
    ap.add_argument("--desc-tol", type=float, default=1e-4)
    ap.add_argument("--score-tol", type=float, default=1e-5, help="|keypoint score| deviation between two fp32 evaluations of the detector head")
    ap.add_argument("--mscore-tol", type=float, default=5e-4, help="|match score| deviation (tests/tolerances.py: 5e-4 for the ill-conditioned seeded "
                                                                   "LightGlue weights at K = 1024, 1e-4 where trained-range logits live)")
    ap.add_argument("--min-matches", type=int, default=0, help="fail unless every pair has at least this many reference matches (guards against a vacuous check)")
    ap.add_argument("--assume-sp", action="append", metavar="KEY=VALUE", help="state a SuperPoint hyper-parameter the graph does not reveal")
    ap.add_argument("--assume-lg", action="append", metavar="KEY=VALUE", help="the same for LightGlue")
    ap.add_argument("--save", metavar="FILE.npz", help="write inputs and every graph output (fixture format of tools/gen_onnx_golden.py)")
    a = ap.parse_args(argv)
    make_session, what = _session_factory(a.backend, a.replay)
    if make_session is None:
        print(f"ort_parity: {what}", file=sys.stderr)
        return 2
    for p in (a.superpoint, a.lightglue):
        if p and not os.path.exists(p):
            print(f"ort_parity: {p} not found", file=sys.stderr)
            return 2
    from rover_slam_amd import onnx_weights, synth, weights as Wt
    from oracle import oracle
    oracle.build()
    print(f"graph execution: {what}")
    H, W = a.height, a.width
    frames, _ = synth.make_frames(a.frames, H, W, seed=a.frame_seed, max_shift=16 if a.shift_step == 8 else 8, shift_step=a.shift_step)
    if a.backend == "replay":
        z = np.load(a.replay)
        frames = z["frames"]
        H, W = frames.shape[1:]
        for key, path in (("sp_onnx_weights_sha256", a.superpoint), ("lg_onnx_weights_sha256", a.lightglue)):
            if path and str(z[key]) != weights_sha256(path):
                print(f"ort_parity: {path} does not hold the weights the recording was made with ({key})", file=sys.stderr)
                return 2
    # the graph's baked-in hyper-parameters first: a deviation below must be arithmetic, not a silently different K / radius / threshold
    read, problems = onnx_weights.read_superpoint_hparams(a.superpoint)
    print(f"{a.superpoint}: hyper-parameters read from the graph: {read}" + (f"; unresolved: {problems}" if problems else ""))
    try:
        wsp, hp = onnx_weights.convert(a.superpoint, 1, onnx_weights._parse_assume(a.assume_sp))
    except ValueError as e:
        print(f"ort_parity: {e}\n  (state what the graph does not reveal with --assume-sp KEY=VALUE)", file=sys.stderr)
        return 2
    sp = make_session(a.superpoint)
    in_names, out_names = [i.name for i in sp.get_inputs()], [o.name for o in sp.get_outputs()]
    if in_names != ["image"] or out_names[:3] != ["keypoints", "scores", "descriptors"]:     # superpoint_onnx.cc:100,133-134 binds these by name
        print(f"ort_parity: {a.superpoint}: inputs {in_names} / outputs {out_names} are not the names the reference binds "
              "(image -> keypoints, scores, descriptors)", file=sys.stderr)
        return 1
    ctx = None
    tmpdir = None
    if a.gpu:
        import tempfile
        from rover_slam_amd import capi
        tmpdir = tempfile.TemporaryDirectory()
        ctx = capi.Context(0)
        if a.gpu_onnx:                                                   # the deployment route without Python: the library reads the graph file itself
            ctx.load_weights(sp_path=a.superpoint)
        else:
            sp_rfew = os.path.join(tmpdir.name, "superpoint.rfew")      # .onnx -> RFEW v2 (onnx_weights.py) -> rfe_load_weights
            Wt.save(sp_rfew, wsp, 1, hp)
            ctx.load_weights(sp_path=sp_rfew)
        got = ctx.get_hparams()
        assert (got["sp_max_keypoints"], got["sp_nms_radius"], got["sp_remove_borders"], got["sp_topk_always"]) == \
               (hp["max_keypoints"], hp["nms_radius"], hp["remove_borders"], hp["topk_always"]), "rfe_load_weights lost the file's hyper-parameters"
    bad = False
    feats, saved = [], {"frames": frames, "sp_hparams": np.array([hp[k] for k in Wt.SP_HPARAMS], np.float64), "recorded_from": what,
                        "sp_inputs": np.array(in_names), "sp_outputs": np.array(out_names), "sp_onnx_weights_sha256": weights_sha256(a.superpoint)}
    for i, img in enumerate(frames):
        x = (img.astype(np.float32) / 255.0)[None, None]                       # NormalizeImage, transform.cpp:3-17
        k_ref, s_ref, d_ref = sp.run(["keypoints", "scores", "descriptors"], {"image": x})
        if k_ref.dtype != np.int64 or k_ref.ndim != 3 or k_ref.shape[0] != 1 or k_ref.shape[2] != 2 or s_ref.shape != k_ref.shape[:2] \
                or d_ref.shape != k_ref.shape[:2] + (256,):                       # superpoint_onnx.cc:169-181 reads int64 [1,K,2], f32 [1,K], f32 [1,K,256]
            print(f"ort_parity: frame {i}: output layout {k_ref.dtype}{k_ref.shape} / {s_ref.shape} / {d_ref.shape} is not int64 [1,K,2] / [1,K] / [1,K,256]", file=sys.stderr)
            return 1
        k_ref, s_ref, d_ref = k_ref[0], s_ref[0], d_ref[0]
        K = k_ref.shape[0]
        saved.update({f"sp{i}_keypoints": k_ref, f"sp{i}_scores": s_ref, f"sp{i}_descriptors": d_ref})
        o = oracle.superpoint(wsp, img, kmax=hp["max_keypoints"], thr=hp["detection_threshold"], nms_radius=hp["nms_radius"],
                              border=hp["remove_borders"], topk_always=bool(hp["topk_always"]))
        cands = [("oracle", o["n"], o["kxy"], o["score"], o["desc"])]
        if ctx is not None:
            n, kxy, sc, de = ctx.extract(img[None], kmax=hp["max_keypoints"], thr=hp["detection_threshold"])
            cands.append(("hip", int(n[0]), kxy[0], sc[0], de[0]))
        for name, n, kxy, sc, de in cands:
            c = compare_keypoints(k_ref, s_ref, d_ref, n, kxy, sc, de, a.score_tol)
            print(f"frame {i} {name}: K_ref={K} K={n} same_set={c['same_set']} same_order={c['same_order']} order_ok={c['order_ok']} "
                  f"max|dscore|={c['dscore']:.3g} max desc L2={c['ddesc']:.3g}")
            bad |= (not c["same_set"]) or (not c["order_ok"]) or c["dscore"] > a.score_tol or c["ddesc"] > a.desc_tol
        feats.append((k_ref, d_ref))
    if a.lightglue:
        read, problems = onnx_weights.read_lightglue_hparams(a.lightglue)
        print(f"{a.lightglue}: hyper-parameters read from the graph: {read}" + (f"; unresolved: {problems}" if problems else ""))
        try:
            wlg, hpl = onnx_weights.convert(a.lightglue, 2, onnx_weights._parse_assume(a.assume_lg))
        except ValueError as e:
            print(f"ort_parity: {e}\n  (state what the graph does not reveal with --assume-lg KEY=VALUE)", file=sys.stderr)
            return 2
        lg = make_session(a.lightglue)
        in_names, out_names = [i.name for i in lg.get_inputs()], [o.name for o in lg.get_outputs()]
        if in_names != ["kpts0", "kpts1", "desc0", "desc1"] or out_names[:2] != ["matches0", "mscores0"]:   # lightglue_onnx.cpp:168-172,210-211
            print(f"ort_parity: {a.lightglue}: inputs {in_names} / outputs {out_names} are not the names the reference binds", file=sys.stderr)
            return 1
        saved.update({"lg_hparams": np.array([hpl[k] for k in Wt.LG_HPARAMS], np.float64), "lg_inputs": np.array(in_names), "lg_outputs": np.array(out_names),
                      "lg_onnx_weights_sha256": weights_sha256(a.lightglue)})
        if ctx is not None:
            if a.gpu_onnx:
                ctx.load_weights(lg_path=a.lightglue)
            else:
                lg_rfew = os.path.join(tmpdir.name, "lightglue_sim.rfew")
                Wt.save(lg_rfew, wlg, 2, hpl)
                ctx.load_weights(lg_path=lg_rfew)
            assert ctx.get_hparams()["lg_filter_threshold"] == np.float32(hpl["filter_threshold"])
        for i in range(len(feats) - 1):
            (k0, d0), (k1, d1) = feats[i], feats[i + 1]
            k0n = oracle.normalize_keypoints(k0.astype(np.float32), H, W)   # NormalizeKeypoints, transform.cpp:19-32
            k1n = oracle.normalize_keypoints(k1.astype(np.float32), H, W)
            m_ref, ms_ref = lg.run(["matches0", "mscores0"], {"kpts0": k0n[None], "kpts1": k1n[None], "desc0": d0[None], "desc1": d1[None]})
            if m_ref.dtype != np.int64 or m_ref.ndim != 2 or m_ref.shape[1] != 2 or ms_ref.shape != m_ref.shape[:1]:   # lightglue_onnx.cpp:404-409
                print(f"ort_parity: pair {i}: output layout {m_ref.dtype}{m_ref.shape} / {ms_ref.shape} is not int64 [S,2] / [S]", file=sys.stderr)
                return 1
            saved.update({f"lg{i}_matches0": m_ref, f"lg{i}_mscores0": ms_ref, f"lg{i}_kpts0": k0n[None], f"lg{i}_kpts1": k1n[None]})
            o = oracle.lightglue(wlg, k0n, k1n, d0, d1, filter_thr=hpl["filter_threshold"])
            cands = [("oracle", o["pairs"], o["ms"])]
            if ctx is not None:
                S, pairs, ms = ctx.match(k0n[None], k1n[None], d0[None], d1[None], [len(k0n)], [len(k1n)], filter_thr=hpl["filter_threshold"])
                cands.append(("hip", pairs[0, :S[0]], ms[0, :S[0]]))
            for name, pairs, ms in cands:
                c = compare_matches(m_ref, ms_ref, pairs, ms, hpl["filter_threshold"], a.mscore_tol)
                print(f"pair {i} {name}: S_ref={len(m_ref)} S={len(pairs)} identical={c['identical']} lists_ok={c['lists_ok']} "
                      f"one_sided={c['one_sided']} max|dmscore|={c['dev']:.3g}")
                bad |= (not c["lists_ok"]) or c["dev"] > a.mscore_tol
            if len(m_ref) < a.min_matches:
                print(f"pair {i}: only {len(m_ref)} reference matches (< --min-matches {a.min_matches}): the comparison is vacuous", file=sys.stderr)
                bad = True
    if a.save:
        np.savez_compressed(a.save, **saved)
        print(f"graph outputs written to {a.save}")
    if ctx is not None:
        ctx.close()
        tmpdir.cleanup()
    print("ort_parity: " + ("DEVIATION" if bad else "within the bars") + f" (desc {a.desc_tol:g}, score {a.score_tol:g}, mscore {a.mscore_tol:g})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
