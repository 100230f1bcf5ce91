"""SURVEY 8(f) N1, validated against files a REAL exporter wrote.

tests/test_onnx_weights.py feeds the converter ONNX bytes produced by its own protobuf writer.  Here the bytes come from
PyTorch's C++ ONNX serialiser (torch.onnx's TorchScript path: `_model_to_graph` + `Graph._export_onnx`; needs neither the `onnx`
package nor onnxruntime): plain-torch modules shaped like the PUBLISHED SuperPoint (magicleap naming conv1a .. convDb, the names
the reference's dead include/SuperPoint.h:24-41 uses) and LightGlue (cvg naming `transformers.{i}.self_attn.Wqkv`, interleaved
qkv rows, `log_assignment.{i}`) are loaded with this repo's synthetic weights, exported with and without constant folding
(folding turns every 3-D Linear into an anonymous, TRANSPOSED `onnx::MatMul_*` constant + a named bias Add; without it the
named weight stays behind a Transpose node) and with the tensor names of the reference's graphs -- `image` ->
`keypoints`/`scores`/`descriptors` (src/Extractors/superpoint_onnx.cc:100,133-134), `kpts0/kpts1/desc0/desc1` ->
`matches0/mscores0` (src/Matchers/lightglue_onnx.cpp:168-172,210-211).  Required: `onnx_weights.convert_*` returns a blob
BIT-IDENTICAL to `weights.make_*`.  The same torch modules also pin the oracle once more: their forward (published formulas,
written here from the papers' reference code) must agree with oracle.lightglue / oracle.superpoint maps.

Nothing here is shipped or read at run time by the product; the .onnx files live in tmp_path only."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from rover_slam_amd import onnx_weights as OW, weights as Wt  # noqa: E402
import onnx_export as X  # noqa: E402  (tools/, on sys.path through conftest.py)
from onnx_export import SuperPointPublished, LightGluePublished  # noqa: E402,F401

_load_sp, _load_lg, _lg_inputs = X.load_sp, X.load_lg, X.lg_inputs


def _serialise(*a, **k):
    """ModelProto bytes from torch's own C++ serialiser, or skip when this torch build lacks the internals."""
    try:
        return X.serialise(*a, **k)
    except X.ExporterUnavailable as e:                       # pragma: no cover
        pytest.skip(str(e))


# ---------------------------------------------------------------------------------------------- published SuperPoint


@pytest.mark.parametrize("fold", [True, False])
def test_superpoint_onnx_from_torch_exporter(tmp_path, fold):
    blob = Wt.make_superpoint(seed=5)
    m = _load_sp(SuperPointPublished(), blob)
    proto = _serialise(m, (torch.rand(1, 1, 32, 40),), ["image"], ["keypoints", "scores", "descriptors"], fold,
                       {"image": {2: "h", 3: "w"}})
    path = tmp_path / "superpoint.onnx"
    path.write_bytes(proto)
    inits, nodes = OW.read_model(str(path))
    assert {"conv1a.weight", "convDb.bias"} <= set(inits) and any(n["op"] == "Conv" for n in nodes)
    assert any("image" in n["inputs"] for n in nodes)
    got = OW.convert_superpoint(str(path))
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), blob.view(np.uint32))   # bit-identical


def test_superpoint_published_forward_matches_oracle(oracle):
    """The published forward (torch CPU) against the oracle's dense maps: pre-NMS score map and normalised descriptor map."""
    from rover_slam_amd import synth
    wsp = Wt.make_superpoint(seed=7)
    m = _load_sp(SuperPointPublished(), wsp)
    img = synth.make_frames(1, 64, 96, seed=3)[0][0]
    with torch.no_grad():
        _, scores, desc = m(torch.from_numpy(img.astype(np.float32) * np.float32(1.0 / 255.0))[None, None])
    r = oracle.superpoint(wsp, img, kmax=16, debug=True)
    assert np.abs(scores[0].numpy() - r["scoremap"]).max() < 2e-5
    assert np.abs(desc[0].permute(1, 2, 0).numpy() - r["descmap"]).max() < 1e-5


@pytest.mark.parametrize("fold", [True, False])
def test_lightglue_onnx_from_torch_exporter(tmp_path, fold):
    blob = Wt.make_lightglue(seed=3)
    m = _load_lg(LightGluePublished(), blob)
    k0, k1, d0, d1 = (torch.from_numpy(a)[None] for a in _lg_inputs(12, 9, 0))
    names = ["kpts0", "kpts1", "desc0", "desc1"]
    proto = _serialise(m, (k0, k1, d0, d1), names, ["matches0", "mscores0"], fold, {n: {1: "n" + n[-1]} for n in names})
    path = tmp_path / "lightglue_sim.onnx"
    path.write_bytes(proto)
    inits, nodes = OW.read_model(str(path))
    if fold:   # what a folded export looks like: anonymous transposed MatMul constants, named biases, no Linear weight by name
        assert any(k.startswith("onnx::MatMul_") for k in inits) and "transformers.0.self_attn.Wqkv.weight" not in inits
        assert "transformers.0.self_attn.Wqkv.bias" in inits
    else:
        assert "transformers.0.self_attn.Wqkv.weight" in inits
    assert not any(k.startswith("log_assignment.0.") for k in inits)          # dead heads never reach the file
    assert {"kpts0", "kpts1", "desc0", "desc1"} <= {i for n in nodes for i in n["inputs"]}
    got = OW.convert_lightglue(str(path))
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), blob.view(np.uint32))   # bit-identical


def test_lightglue_exporter_file_with_wrong_depth_is_refused(tmp_path):
    """a 3-layer model where 9 are expected: the converter lists what it cannot place instead of guessing"""
    m = LightGluePublished(n_layers=3).eval()
    k0, k1, d0, d1 = (torch.from_numpy(a)[None] for a in _lg_inputs(8, 8, 1))
    proto = _serialise(m, (k0, k1, d0, d1), ["kpts0", "kpts1", "desc0", "desc1"], ["matches0", "mscores0"], True)
    path = tmp_path / "short.onnx"
    path.write_bytes(proto)
    with pytest.raises(ValueError, match="cannot place LightGlue tensors"):
        OW.convert_lightglue(str(path))


def test_lightglue_published_forward_matches_oracle(oracle):
    """The published forward (interleaved Wqkv, rotate_half rotary, [x | msg] FFN, dual log-softmax assignment) in torch against
    oracle.lightglue on the same weights: a third, independent pin of the oracle's restatement (besides the HF fixtures and
    the float64 numpy graph of tools/lg_tolerance_study.py), and of the q|k|v de-interleaving the converter performs."""
    blob = Wt.make_lightglue(seed=11)
    m = _load_lg(LightGluePublished(), blob)
    k0, k1, d0, d1 = _lg_inputs(48, 40, 2)
    with torch.no_grad():
        pairs, ms, x0, x1, scores = m(*(torch.from_numpy(a)[None] for a in (k0, k1, d0, d1)))
    r = oracle.lightglue(blob, k0, k1, d0, d1, debug=True)
    assert np.abs(x0[0].numpy() - r["x0"]).max() < 1e-4 and np.abs(x1[0].numpy() - r["x1"]).max() < 1e-4
    assert np.abs(scores[0].numpy() - r["scores"]).max() < 1e-5 * np.abs(r["scores"]).max()
    assert r["S"] > 10 and np.array_equal(pairs.numpy().astype(np.int32), r["pairs"])
    assert np.abs(ms.numpy() - r["ms"]).max() < 1e-4
