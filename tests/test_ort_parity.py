"""tools/ort_parity.py executed END TO END: the oracle against an execution of the exported GRAPH FILES.

The reference's arithmetic is `Session::Run` on onnxmodel/superpoint.onnx / lightglue_sim.onnx
(src/Extractors/superpoint_onnx.cc:133-136, src/Matchers/lightglue_onnx.cpp:210-214); both blobs and onnxruntime are absent from this
image.  Here the published SuperPoint WITH ITS REAL TAIL and the fused LightGlue are exported by torch's own ONNX serialiser at two
hyper-parameter settings, the FILES are executed node by node by tools/mini_onnx.py, and the harness must exit 0: same keypoints
in the same order (up to fp32 near-ties), int64 [1,K,2] layout and (x, y) order, scores, descriptors <= 1e-4, matches0 / mscores0 --
through weights and hyper-parameters that `onnx_weights.convert` read back from the same file.  The run is recorded and must equal the
committed fixture tests/golden/onnx_s*.npz that the `-m gpu` test replays against librover_fe.so."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

import gen_onnx_golden as G  # noqa: E402  (tools/)
import onnx_export as X  # noqa: E402
import ort_parity as P  # noqa: E402
import mini_onnx as M  # noqa: E402


def _export(tmp_path, case):
    try:
        return G.export_case(str(tmp_path), case)
    except X.ExporterUnavailable as e:                         # pragma: no cover
        pytest.skip(str(e))


@pytest.mark.parametrize("case", ["s0", "s1"])
def test_harness_runs_end_to_end_on_graph_execution(tmp_path, golden_dir, case, capsys):
    sp, lg = _export(tmp_path, case)
    rec = str(tmp_path / "rec.npz")
    rc = P.main(G.harness_args(case, sp, lg) + ["--backend", "mini", "--save", rec])
    out = capsys.readouterr().out
    assert rc == 0, out
    assert "same_set=True" in out and "same_set=False" not in out and "identical=True" in out and "within the bars" in out
    # the recording equals the committed fixture (integers exactly, floats to the run-to-run noise of multi-threaded torch-CPU kernels)
    a, b = np.load(rec), np.load(os.path.join(golden_dir, f"onnx_{case}.npz"))
    assert sorted(a.files) == sorted(b.files)
    assert str(a["sp_onnx_weights_sha256"]) == str(b["sp_onnx_weights_sha256"]) and str(a["lg_onnx_weights_sha256"]) == str(b["lg_onnx_weights_sha256"])
    for k in a.files:
        if a[k].dtype.kind in "iu":
            assert np.array_equal(a[k], b[k]), k
        elif a[k].dtype.kind == "f":
            assert a[k].shape == b[k].shape and np.abs(a[k] - b[k]).max() <= 2e-6, k
    # the replay backend (what the GPU box runs) accepts the committed fixture and reaches the same verdict on the oracle
    assert P.main(G.harness_args(case, sp, lg) + ["--backend", "replay", "--replay", os.path.join(golden_dir, f"onnx_{case}.npz")]) == 0


def test_harness_at_the_benchmark_size(tmp_path, capsys):
    """the same, live, at the size bench.py runs: two 480 x 640 frames of its stream (cell-aligned shifts), K = 1024 through the top-k cut, the calibrated
    LightGlue weights -- oracle against graph execution at 1e-4 with > 100 matches (no fixture: 2 x 1 MB of descriptors per frame)"""
    sp, lg = _export(tmp_path, "s0")
    rc = P.main(["--superpoint", sp, "--lightglue", lg, "--backend", "mini", "--frames", "2", "--height", "480", "--width", "640", "--shift-step", "8",
                 "--mscore-tol", "1e-4", "--min-matches", "100"])
    out = capsys.readouterr().out
    assert rc == 0 and "K_ref=1024 K=1024 same_set=True" in out and "identical=True" in out, out


def test_replay_refuses_other_weights(tmp_path, golden_dir):
    """a graph file holding other weights than the recording's is not silently compared"""
    try:
        sp, _ = X.export_sp(str(tmp_path), X.SETTINGS[0], seed=8)
        lg, _ = X.export_lg(str(tmp_path), 0.1, seed=11, calibrated=True)
    except X.ExporterUnavailable as e:                         # pragma: no cover
        pytest.skip(str(e))
    assert P.main(G.harness_args("s0", sp, lg) + ["--backend", "replay", "--replay", os.path.join(golden_dir, "onnx_s0.npz")]) == 2


def test_harness_detects_a_deviation(tmp_path, capsys):
    """the bars bite: a tolerance no two fp32 evaluations can meet -> exit 1, named as a deviation"""
    sp, lg = _export(tmp_path, "s1")
    args = G.harness_args("s1", sp, lg)
    assert P.main(args + ["--backend", "mini", "--score-tol", "1e-9"]) == 1
    assert "DEVIATION" in capsys.readouterr().out


def test_ort_backend_reports_missing_prerequisites(tmp_path):
    try:
        import onnxruntime  # noqa: F401
        pytest.skip("onnxruntime is installed here")
    except ImportError:
        pass
    assert P.main(["--superpoint", str(tmp_path / "x.onnx")]) == 2


# ---------------------------------------------------------------------------------------------- the interpreter itself
def test_mini_onnx_equals_the_torch_modules_at_another_size(tmp_path):
    """graph execution == module execution at a size OTHER than the trace size (this is what exposed a trace-time constant in an
    earlier export: descriptors off by 0.25 through a frozen image size), incl. int64 (x, y) keypoints and the match filter"""
    from rover_slam_amd import synth, weights as Wt
    sp, blob = X.export_sp(str(tmp_path), X.SETTINGS[0], seed=5)
    img = synth.make_frames(1, 120, 160, seed=3)[0][0]
    x = (img.astype(np.float32) / 255.0)[None, None]
    k, s, d = M.InferenceSession(sp).run(["keypoints", "scores", "descriptors"], {"image": x})
    with torch.no_grad():
        k2, s2, d2 = X.load_sp(X.SuperPointWithTail(X.SETTINGS[0]), blob)(torch.from_numpy(x))
    assert k.dtype == np.int64 and np.array_equal(k, k2.numpy()) and np.array_equal(s, s2.numpy()) and np.abs(d - d2.numpy()).max() < 1e-6
    lg, wlg = X.export_lg(str(tmp_path), 0.1, seed=3)
    k0, k1, d0, d1 = X.lg_inputs(48, 40, 2)
    m, ms = M.InferenceSession(lg).run(["matches0", "mscores0"], {"kpts0": k0[None], "kpts1": k1[None], "desc0": d0[None], "desc1": d1[None]})
    with torch.no_grad():
        p, s3, *_ = X.load_lg(X.LightGluePublished(), wlg)(*(torch.from_numpy(a)[None] for a in (k0, k1, d0, d1)))
    assert m.dtype == np.int64 and len(m) > 10 and np.array_equal(m, p.numpy()) and np.abs(ms - s3.numpy()).max() < 1e-6


def test_mini_onnx_operator_semantics():
    """the operators whose ONNX semantics differ from the obvious numpy call"""
    n = lambda op, **attrs: dict(op=op, attrs=attrs, name="t", inputs=[], outputs=["y"])
    x = np.array([0.5, 0.9, 0.9, 0.1, 0.9], np.float32)
    v, i = M._topk(n("TopK", axis=0), [x, np.array([3])])
    assert i.tolist() == [1, 2, 4] and v.tolist() == [np.float32(0.9)] * 3                      # ties: lower index first
    assert M._int_div(np.array([-7, 7]), np.array([2, -2])).tolist() == [-3, -3]                  # truncation, not floor
    a = np.arange(10)
    assert M._slice(n("Slice"), [a, np.array([-3]), np.array([2 ** 62]), np.array([0]), np.array([1])])[0].tolist() == [7, 8, 9]
    assert M._slice(n("Slice"), [a, np.array([8]), np.array([-(2 ** 62)]), np.array([0]), np.array([-3])])[0].tolist() == [8, 5, 2]
    d = np.zeros((3, 4), np.float32)
    out = M._scatter_nd(n("ScatterND"), [d, np.array([[0], [2]]), -np.ones((2, 4), np.float32)])[0]
    assert out[0].tolist() == [-1] * 4 and out[1].tolist() == [0] * 4 and d.sum() == 0           # copy, not in place
    assert M._reshape(n("Reshape"), [np.zeros((2, 3, 4)), np.array([0, -1])])[0].shape == (2, 12)
    assert M._OPS["NonZero"](n("NonZero"), [np.array([[0, 1], [1, 0]])])[0].tolist() == [[0, 1], [1, 0]]
    assert M._unsqueeze(n("Unsqueeze"), [np.zeros((3,)), np.array([0, 2])])[0].shape == (1, 3, 1)
    with pytest.raises(NotImplementedError, match="operators not implemented"):
        class S(M.InferenceSession):
            def __init__(self):
                pass
        s = S(); s.path = "x"
        M.OW.read_model, keep = (lambda p: ({}, [dict(op="FancyOp", inputs=[], outputs=[], name="", attrs={})])), M.OW.read_model
        M.OW.read_graph_io, keep2 = (lambda p: dict(inputs=[], outputs=[], opset=17)), M.OW.read_graph_io
        try:
            M.InferenceSession("x")
        finally:
            M.OW.read_model, M.OW.read_graph_io = keep, keep2
