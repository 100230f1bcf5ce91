"""`.onnx` in, from C++: rover-slam_amd/csrc/onnx_load.hip against its specification rover-slam_amd/onnx_weights.py.

The reference builds its sessions from onnxmodel/superpoint.onnx (src/Extractors/SPextractor.cc:92-94) and the hard-coded
onnxmodel/lightglue_sim.onnx (src/Matchers/lightglue_onnx.cpp:38).  rfe_load_weights now takes such files directly; the reader is exercised
here WITHOUT a GPU through rfe_k_onnx_convert on every kind of graph file the Python tests use -- files written by torch's own ONNX
serialiser (published SuperPoint with its real tail at two settings and with a constant top-k, fused LightGlue with and without constant
folding, at two filter thresholds) and files written by tests/test_onnx_weights.py's protobuf writer (anonymous, onnx-simplifier-style
graphs: Gemm with transB 0 / 1, MatMul + Add, decomposed LayerNorm) -- and must give the SAME blob, bit for bit, the same hyper-parameters,
and refuse what the Python converter refuses.  The `-m gpu` test loads the .onnx through rfe_load_weights and extracts / matches."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from rover_slam_amd import capi, onnx_weights as OW, weights as Wt  # noqa: E402
import onnx_export as X  # noqa: E402
import test_onnx_weights as TW  # noqa: E402


def cpp_convert(path, kind, weights_only=False):
    """-> (blob, hparams dict of that kind) or raises ValueError(reason)"""
    blob = np.empty(Wt.SP_COUNT if kind == 1 else Wt.LG_COUNT, np.float32)
    hp = capi.HParams()
    err = C.create_string_buffer(4096)
    rc = capi.lib.rfe_k_onnx_convert(str(path).encode(), kind, int(weights_only), blob.ctypes.data_as(C.POINTER(C.c_float)), C.byref(hp), err, 4096)
    if rc != 0:
        raise ValueError(err.value.decode(errors="replace"))
    d = hp.as_dict()
    if kind == 1:
        return blob, {"max_keypoints": d["sp_max_keypoints"], "detection_threshold": d["sp_detection_threshold"], "nms_radius": d["sp_nms_radius"],
                      "remove_borders": d["sp_remove_borders"], "topk_always": d["sp_topk_always"]}
    return blob, {"layers": d["lg_layers"], "heads": d["lg_heads"], "filter_threshold": d["lg_filter_threshold"]}


def _same(path, kind):
    pb, php = OW.convert(str(path), kind)
    cb, chp = cpp_convert(path, kind)
    assert np.array_equal(cb.view(np.uint32), pb.view(np.uint32)), "C++ blob differs from the Python converter's"
    assert set(chp) == set(php)
    for k in php:
        assert np.float32(chp[k]) == np.float32(php[k]), (k, chp[k], php[k])
    return cb, chp


def _export(fn, *a, **k):
    try:
        return fn(*a, **k)
    except X.ExporterUnavailable as e:                         # pragma: no cover
        pytest.skip(str(e))


@pytest.mark.parametrize("setting", [0, 1])
def test_superpoint_exported_graph_bit_identical_to_python(tmp_path, setting):
    path, blob = _export(X.export_sp, tmp_path, X.SETTINGS[setting], seed=5)
    cb, hp = _same(path, 1)
    assert np.array_equal(cb, blob)
    assert (hp["max_keypoints"], hp["nms_radius"], hp["remove_borders"], hp["topk_always"]) == \
           (X.SETTINGS[setting]["max_keypoints"], X.SETTINGS[setting]["nms_radius"], X.SETTINGS[setting]["remove_borders"], 1)


def test_superpoint_constant_topk_and_tail_free_graph(tmp_path):
    path, _ = _export(X.export_sp, tmp_path, dict(X.SETTINGS[0], max_keypoints=16), topk="const")
    _, hp = _same(path, 1)
    assert hp["max_keypoints"] == 16 and hp["topk_always"] == 0
    # no readable tail: refused like the Python converter (same keys named, in the same order), the weights alone still convert
    m = X.load_sp(X.SuperPointPublished(), Wt.make_superpoint(seed=5))
    proto = _export(X.serialise, m, (torch.rand(1, 1, 32, 40),), ["image"], ["keypoints", "scores", "descriptors"], True, {"image": {2: "h", 3: "w"}})
    p2 = tmp_path / "notail.onnx"
    p2.write_bytes(proto)
    with pytest.raises(ValueError, match="graph hyper-parameters refused.*nms_radius.*max_keypoints.*remove_borders.*grid_sample"):
        cpp_convert(p2, 1)
    with pytest.raises(ValueError, match="graph hyper-parameters refused"):
        OW.convert(str(p2), 1)
    wb, _ = cpp_convert(p2, 1, weights_only=True)
    assert np.array_equal(wb, OW.convert_superpoint(str(p2)))


@pytest.mark.parametrize("fold,thr,calibrated", [(True, 0.1, False), (False, 0.25, True)])
def test_lightglue_exported_graph_bit_identical_to_python(tmp_path, fold, thr, calibrated):
    path, blob = _export(X.export_lg, tmp_path, thr, seed=3, fold=fold, calibrated=calibrated)
    cb, hp = _same(path, 2)
    assert np.array_equal(cb, blob) and hp["layers"] == 9 and hp["heads"] == 4 and hp["filter_threshold"] == np.float32(thr)


def test_lightglue_other_depth_is_refused_with_its_depth_named(tmp_path):
    path, _ = _export(X.export_lg, tmp_path, 0.1, n_layers=3)
    with pytest.raises(ValueError, match="3 layers of 4 heads"):
        cpp_convert(path, 2)
    with pytest.raises(ValueError, match="cannot place LightGlue tensors"):
        cpp_convert(path, 2, weights_only=True)


def test_anonymous_graphs_of_the_python_tests(tmp_path):
    """the hand-written protobuf files of tests/test_onnx_weights.py (no parameter names: Linear layers by order of first use, Gemm transB 0 / 1,
    MatMul + Add, decomposed LayerNorm; SuperPoint by shape and file order): same blobs, same refusals"""
    cases = TW.anonymous_cases(tmp_path)
    assert len(cases) == 8
    for path, kind, ok in cases:
        if ok:
            want = OW.convert_superpoint(path) if kind == 1 else OW.convert_lightglue(path)
            got, _ = cpp_convert(path, kind, weights_only=True)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), path
        else:
            with pytest.raises(ValueError) as pe:
                (OW.convert_superpoint if kind == 1 else OW.convert_lightglue)(path)
            with pytest.raises(ValueError) as ce:
                cpp_convert(path, kind, weights_only=True)
            # the same diagnosis: what could not be placed, with position and shapes where the Python converter gives them
            for needle in ("cannot place SuperPoint tensors", "cannot place LightGlue tensors", "83 Linear layers in the graph", "has 84",
                           "expected layers.3.cross.Wo (256, 256) + bias, found (512, 512)", "Linear #0: expected posenc.Wr (32, 2), found (256, 256)"):
                assert (needle in str(pe.value)) == (needle in str(ce.value)), (path, needle, str(ce.value))


def test_not_a_model_file(tmp_path):
    p = tmp_path / "junk.onnx"
    p.write_bytes(b"\xff\xff\xff\xffthis is not protobuf")
    with pytest.raises(ValueError, match="not an ONNX ModelProto"):
        cpp_convert(p, 1)
    with pytest.raises(ValueError, match="cannot open"):
        cpp_convert(tmp_path / "absent.onnx", 2)


@pytest.mark.gpu
def test_rfe_load_weights_takes_onnx_files(tmp_path, oracle):
    """the deployment route without Python: rfe_load_weights(superpoint.onnx, lightglue_sim.onnx) -> hyper-parameters applied, extraction and
    matching equal to the oracle run on the converter's blob"""
    from rover_slam_amd import synth
    sp, wsp = _export(X.export_sp, tmp_path, X.SETTINGS[1], seed=7, desc_center="auto")
    lg, wlg = _export(X.export_lg, tmp_path, 0.25, seed=11)     # the seeded law: 40 matches on these small frames (the calibrated one leaves 1 above 0.25)
    ctx = capi.Context(0)
    try:
        ctx.load_weights(sp_path=sp, lg_path=lg)
        hp = ctx.get_hparams()
        assert (hp["sp_max_keypoints"], hp["sp_nms_radius"], hp["sp_remove_borders"], hp["sp_topk_always"]) == (2048, 3, 2, 1)
        assert hp["sp_detection_threshold"] == np.float32(0.005) and hp["lg_filter_threshold"] == np.float32(0.25)
        frames, _ = synth.make_frames(2, 120, 160, seed=20240314, max_shift=16, shift_step=8)
        n, kxy, score, desc = ctx.extract(frames, kmax=2048, thr=hp["sp_detection_threshold"])
        feats = []
        for b in range(2):
            r = oracle.superpoint(wsp, frames[b], kmax=2048, thr=0.005, nms_radius=3, border=2, topk_always=True)
            assert n[b] == r["n"] and np.array_equal(kxy[b], r["kxy"]) and np.array_equal(score[b], r["score"]) and np.array_equal(desc[b], r["desc"])
            feats.append(r)
        k0 = oracle.normalize_keypoints(feats[0]["kxy"][:n[0]].astype(np.float32), 120, 160)
        k1 = oracle.normalize_keypoints(feats[1]["kxy"][:n[1]].astype(np.float32), 120, 160)
        S, pairs, ms = ctx.match(k0[None], k1[None], desc[0, :n[0]][None], desc[1, :n[1]][None], [int(n[0])], [int(n[1])], filter_thr=hp["lg_filter_threshold"])
        r = oracle.lightglue(wlg, k0, k1, desc[0, :n[0]], desc[1, :n[1]], filter_thr=0.25)
        from tolerances import LG_SCORE_TOL
        assert r["S"] > 10 and S[0] == r["S"] and np.array_equal(pairs[0, :S[0]], r["pairs"]) and np.abs(ms[0, :S[0]] - r["ms"]).max() < LG_SCORE_TOL
        (tmp_path / "short").mkdir()
        short, _ = X.export_lg(str(tmp_path / "short"), 0.1, n_layers=3)
        with pytest.raises(capi.RfeError, match="3 layers of 4 heads"):
            ctx.load_weights(lg_path=short)
        # the ctx keeps its previous LightGlue after a refused load
        S2, _, _ = ctx.match(k0[None], k1[None], desc[0, :n[0]][None], desc[1, :n[1]][None], [int(n[0])], [int(n[1])], filter_thr=0.25)
        assert S2[0] == S[0]
    finally:
        ctx.close()


def test_damaged_files_are_refused_not_crashed_on(tmp_path):
    """truncations, random byte flips and insertions in an exported graph file: the C++ reader either converts (a flip inside a weight payload) or refuses with a
    reason -- it never reads past a tensor its dims lie about (an element count that is not the product of the dims drops the tensor)"""
    sp, _ = _export(X.export_sp, tmp_path, X.SETTINGS[0], seed=5)
    data = bytearray(open(sp, "rb").read())
    rng = np.random.default_rng(0)
    p = tmp_path / "m.onnx"
    refused = 0
    for it in range(80):
        d = bytearray(data)
        if it % 4 == 0:
            d = d[:int(rng.integers(1, len(d)))]
        elif it % 4 == 1:
            for _ in range(int(rng.integers(1, 20))):
                d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
        elif it % 4 == 2:
            pos = int(rng.integers(0, min(len(d), 200000)))
            d[pos:pos + 8] = bytes(rng.integers(0, 256, 8, dtype=np.uint8))
        else:
            pos = int(rng.integers(0, len(d)))
            d = d[:pos] + bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8)) + d[pos:]
        p.write_bytes(bytes(d))
        try:
            cpp_convert(p, 1)
        except ValueError as e:
            refused += 1
            assert str(e)                       # a reason, always
    assert refused >= 20
