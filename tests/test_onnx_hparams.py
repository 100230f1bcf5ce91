"""The hyper-parameters the reference's graphs carry.

The reference's C++ holds none of them: `SuperPointOnnxRunner::Extractor_PostProcess` takes K from the SHAPE of the `keypoints`
output (src/Extractors/superpoint_onnx.cc:169-181) and `Matcher_PostProcess_fused` consumes matches0 / mscores0 as they come
(src/Matchers/lightglue_onnx.cpp:404-409) because max_num_keypoints, the detection threshold, the NMS radius, the border,
LightGlue's depth / heads / filter threshold are constants of onnxmodel/superpoint.onnx / lightglue_sim.onnx (both missing here).
These tests export the PUBLISHED SuperPoint WITH ITS REAL TAIL (simple_nms = 5 x max_pool2d(2r+1, 1, r), border = -1, threshold,
top-k, grid_sample(bilinear, align_corners) + normalisation) and the published LightGlue with its filter through PyTorch's own C++
ONNX serialiser at two settings, and require
  * `onnx_weights.read_*_hparams` to read every value back from the graph,
  * `onnx_weights.convert` to refuse a graph whose values it cannot read unless the caller states them,
  * the RFEW v2 container to carry them (weights.save / load),
  * the torch tail, fed the ORACLE's dense maps, to reproduce the oracle's keypoints / scores / descriptors at both settings --
    which pins the oracle's NMS radius / border / threshold / top-k rule / sampling against the published code,
  * (-m gpu) librover_fe.so loaded from the v2 file to change its output exactly as the oracle says.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
F = torch.nn.functional

from rover_slam_amd import onnx_weights as OW, weights as Wt, synth  # noqa: E402
from test_onnx_exporter import SuperPointPublished, LightGluePublished, _serialise, _load_sp, _load_lg, _lg_inputs  # noqa: E402
import onnx_export as X  # noqa: E402
from onnx_export import SETTINGS, sp_tail  # noqa: E402


def _export_sp(tmp_path, *a, **k):
    try:
        return X.export_sp(tmp_path, *a, **k)
    except X.ExporterUnavailable as e:                       # pragma: no cover
        pytest.skip(str(e))



@pytest.mark.parametrize("hp", SETTINGS)
def test_superpoint_hparams_read_back_from_exported_graph(tmp_path, hp):
    path, blob = _export_sp(tmp_path, hp)
    read, problems = OW.read_superpoint_hparams(path)
    assert problems == []
    assert read == dict(hp, detection_threshold=float(np.float32(hp["detection_threshold"])), topk_always=1)
    got, hp2 = OW.convert(path, 1)                                                          # weights AND hyper-parameters
    assert np.array_equal(got.view(np.uint32), blob.view(np.uint32)) and hp2 == read
    Wt.save(str(tmp_path / "sp.rfew"), got, 1, hp2)
    blob2, kind, hp3 = Wt.load(str(tmp_path / "sp.rfew"), with_hparams=True)
    assert kind == 1 and np.array_equal(blob2, blob) and hp3 == hp2
    # a stated value that contradicts the graph is refused, an agreeing one accepted
    with pytest.raises(ValueError, match="nms_radius: the graph says"):
        OW.convert(path, 1, assume={"nms_radius": hp["nms_radius"] + 1})
    assert OW.convert(path, 1, assume={"nms_radius": hp["nms_radius"]})[1] == read
    OW.main(["--superpoint", path, "--out-dir", str(tmp_path)])                             # the CLI writes the v2 container
    assert Wt.load(str(tmp_path / "superpoint.rfew"), with_hparams=True)[2] == read


def test_superpoint_constant_topk_is_the_published_rule(tmp_path):
    """torch.topk(scores, k) with a constant k (what tracing the published top_k_keypoints leaves when more than k candidates exist):
    TopK without a Min in front -> topk_always = 0"""
    hp = dict(SETTINGS[0], max_keypoints=16)
    path, _ = _export_sp(tmp_path, hp, topk="const")
    read, problems = OW.read_superpoint_hparams(path)
    assert problems == [] and read["max_keypoints"] == 16 and read["topk_always"] == 0


def test_graph_without_readable_tail_is_refused_unless_stated(tmp_path):
    """test_onnx_exporter's SuperPointPublished replaces the tail by nonzero(scores > 0.5): no NMS pools, no TopK, no border, no
    GridSample.  convert refuses it and names what it could not read; stating the four values is not enough either, because the
    descriptor sampling is structurally absent."""
    blob = Wt.make_superpoint(seed=5)
    m = _load_sp(SuperPointPublished(), blob)
    proto = _serialise(m, (torch.rand(1, 1, 32, 40),), ["image"], ["keypoints", "scores", "descriptors"], True, {"image": {2: "h", 3: "w"}})
    path = tmp_path / "superpoint.onnx"
    path.write_bytes(proto)
    read, problems = OW.read_superpoint_hparams(str(path))
    assert read["nms_radius"] is None and read["max_keypoints"] is None and read["remove_borders"] is None
    assert read["detection_threshold"] == 0.5                     # the one Greater of that graph: read, and visibly not 0.0005
    with pytest.raises(ValueError, match="nms_radius.*max_keypoints.*remove_borders.*grid_sample"):
        OW.convert(str(path), 1)
    with pytest.raises(ValueError, match="grid_sample"):
        OW.convert(str(path), 1, assume={"nms_radius": 4, "max_keypoints": 1024, "remove_borders": 4})
    assert np.array_equal(OW.convert_superpoint(str(path)), blob)  # the weights alone still convert (tools that only need them)


@pytest.mark.parametrize("hp", SETTINGS)
@pytest.mark.parametrize("seed,dustbin", [(7, 0.0), (7, 9.5)])
def test_published_tail_on_oracle_maps_reproduces_oracle(oracle, hp, seed, dustbin):
    """Pins the oracle's tail against the published code at both settings: the torch tail, fed the oracle's own dense maps (so that
    the equality-based NMS sees the same floats), must give the oracle's keypoints and scores exactly and its descriptors to 1e-6 --
    with more candidates than K (top-k cut) and with fewer (dustbin weights: topk_always orders them, the published rule does not)."""
    wsp = Wt.make_superpoint(seed=seed, dustbin_bias=dustbin)
    img = synth.make_frames(1, 120, 160, seed=3)[0][0]
    kmax = hp["max_keypoints"] if dustbin else 150
    h = dict(hp, max_keypoints=kmax)
    r = oracle.superpoint(wsp, img, kmax=kmax, thr=h["detection_threshold"], nms_radius=h["nms_radius"], border=h["remove_borders"],
                          debug=True, topk_always=True)
    with torch.no_grad():
        kp, sc, d = sp_tail(torch.from_numpy(r["scoremap"].copy())[None], torch.from_numpy(r["descmap"].copy()).permute(2, 0, 1)[None], h)
    n = r["n"]
    assert n == kp.shape[0] and (n == kmax) == (not dustbin) and n > 20
    assert np.array_equal(kp.numpy().astype(np.int32), r["kxy"][:n]) and np.array_equal(sc.numpy(), r["score"][:n])
    assert np.abs(d.numpy() - r["desc"][:n]).max() < 1e-6
    if dustbin:      # the published rule leaves the same keypoints in row-major order
        r0 = oracle.superpoint(wsp, img, kmax=kmax, thr=h["detection_threshold"], nms_radius=h["nms_radius"], border=h["remove_borders"])
        assert r0["n"] == n and not np.array_equal(r0["kxy"][:n], r["kxy"][:n])
        lin = lambda k: k[:, 1].astype(np.int64) * 4096 + k[:, 0]
        assert np.array_equal(np.sort(lin(r0["kxy"][:n])), np.sort(lin(r["kxy"][:n]))) and np.all(np.diff(lin(r0["kxy"][:n])) > 0)


def test_settings_change_the_oracle_output(oracle):
    wsp = Wt.make_superpoint(seed=7)
    img = synth.make_frames(1, 120, 160, seed=3)[0][0]
    a, b = (oracle.superpoint(wsp, img, kmax=2048, thr=h["detection_threshold"], nms_radius=h["nms_radius"], border=h["remove_borders"])
            for h in SETTINGS)
    assert a["n"] != b["n"]                        # radius 3 / border 2 keeps more keypoints than radius 4 / border 4
    assert b["kxy"][:b["n"]].min() >= 2 and b["kxy"][:b["n"]].min() < 4


# ---------------------------------------------------------------------------------------------------------------- LightGlue
@pytest.mark.parametrize("thr", [0.1, 0.25])
def test_lightglue_hparams_read_back_from_exported_graph(tmp_path, thr):
    blob = Wt.make_lightglue(seed=3)
    m = _load_lg(LightGluePublished(filter_threshold=thr), blob)
    k0, k1, d0, d1 = (torch.from_numpy(a)[None] for a in _lg_inputs(12, 9, 0))
    names = ["kpts0", "kpts1", "desc0", "desc1"]
    proto = _serialise(m, (k0, k1, d0, d1), names, ["matches0", "mscores0"], True, {n: {1: "n" + n[-1]} for n in names})
    path = tmp_path / "lightglue_sim.onnx"
    path.write_bytes(proto)
    read, problems = OW.read_lightglue_hparams(str(path))
    assert problems == [] and read == {"layers": 9, "heads": 4, "filter_threshold": float(np.float32(thr))}
    got, hp = OW.convert(str(path), 2)
    assert np.array_equal(got.view(np.uint32), blob.view(np.uint32)) and hp == read
    Wt.save(str(tmp_path / "lg.rfew"), got, 2, hp)
    assert Wt.load(str(tmp_path / "lg.rfew"), with_hparams=True)[2] == read


def test_lightglue_other_depth_is_refused_with_its_depth_named(tmp_path):
    m = LightGluePublished(n_layers=3).eval()
    k0, k1, d0, d1 = (torch.from_numpy(a)[None] for a in _lg_inputs(8, 8, 1))
    proto = _serialise(m, (k0, k1, d0, d1), ["kpts0", "kpts1", "desc0", "desc1"], ["matches0", "mscores0"], True)
    path = tmp_path / "short.onnx"
    path.write_bytes(proto)
    read, problems = OW.read_lightglue_hparams(str(path))
    assert read["layers"] == 3 and read["heads"] == 4
    with pytest.raises(ValueError, match="3 layers of 4 heads"):
        OW.convert(str(path), 2)


def test_lightglue_control_flow_and_confidence_heads_are_reported(tmp_path):
    """an early-exit export keeps If / Loop nodes and the per-layer token-confidence Linears: both are named as problems"""
    from test_onnx_weights import _model, _node
    rng = np.random.default_rng(0)
    inits = [(f"w{i}", rng.standard_normal((256, 1)).astype(np.float32)) for i in range(3)] + [(f"b{i}", np.zeros(1, np.float32)) for i in range(3)]
    nodes = [_node("MatMul", ["x", f"w{i}"], [f"m{i}"]) for i in range(3)] + [_node("Add", [f"m{i}", f"b{i}"], [f"c{i}"]) for i in range(3)]
    nodes.append(_node("If", ["cond"], ["y"]))
    path = tmp_path / "early_exit.onnx"
    path.write_bytes(_model(inits, nodes))
    _, problems = OW.read_lightglue_hparams(str(path))
    text = " | ".join(problems)
    assert "control flow ['If']" in text and "3 Linear(256 -> 1) heads" in text


# ---------------------------------------------------------------------------------------------------------------- RFEW container
def test_rfew_v1_files_still_load_with_the_published_defaults(tmp_path):
    blob = Wt.make_superpoint(seed=1)
    Wt.save(str(tmp_path / "v1.rfew"), blob, 1, version=1)
    b, kind, hp = Wt.load(str(tmp_path / "v1.rfew"), with_hparams=True)
    assert kind == 1 and np.array_equal(b, blob) and hp == Wt.SP_HPARAMS
    assert (tmp_path / "v1.rfew").stat().st_size == 20 + 4 * blob.size


# ---------------------------------------------------------------------------------------------------------------- the HIP library
@pytest.mark.gpu
def test_hip_library_applies_the_file_hparams_exactly_as_the_oracle(tmp_path, oracle):
    """load the second setting from an RFEW v2 file -> rfe_get_hparams reports it, and extraction with the
    file's K / threshold (what the C++ shims pass) changes exactly as the oracle says: NMS radius 3, border 2, threshold 0.005,
    K = 2048, top-k always."""
    from rover_slam_amd import capi
    wsp = Wt.make_superpoint(seed=7)
    frames, _ = synth.make_frames(2, 240, 320, seed=9)
    ctx = capi.Context(0)
    outs = []
    for i, hp in enumerate([dict(SETTINGS[0], topk_always=0), dict(SETTINGS[1], topk_always=1)]):
        path = str(tmp_path / f"sp{i}.rfew")
        Wt.save(path, wsp, 1, hp)
        ctx.load_weights(sp_path=path)
        got = ctx.get_hparams()
        assert (got["sp_max_keypoints"], got["sp_nms_radius"], got["sp_remove_borders"], got["sp_topk_always"]) == \
               (hp["max_keypoints"], hp["nms_radius"], hp["remove_borders"], hp["topk_always"])
        assert got["sp_detection_threshold"] == np.float32(hp["detection_threshold"])
        n, kxy, score, desc = ctx.extract(frames, kmax=got["sp_max_keypoints"], thr=got["sp_detection_threshold"])
        for b in range(2):
            r = oracle.superpoint(wsp, frames[b], kmax=hp["max_keypoints"], thr=hp["detection_threshold"], nms_radius=hp["nms_radius"],
                                  border=hp["remove_borders"], topk_always=bool(hp["topk_always"]))
            assert n[b] == r["n"] and np.array_equal(kxy[b], r["kxy"]) and np.array_equal(score[b], r["score"])
            assert np.array_equal(desc[b], r["desc"])
        outs.append((int(n[0]), kxy[0].copy()))
    assert outs[0][0] != outs[1][0]                                    # the two settings really differ on these frames
    # other radii through the run-time kernel, K < Kmax with the unconditional top-k (dustbin weights), no border
    wd = Wt.make_superpoint(seed=7, dustbin_bias=9.5)
    ctx.set_weights(capi.KIND_SUPERPOINT, wd)
    below = 0
    for radius, border, always in ((1, 0, 1), (2, 7, 0), (6, 3, 1), (8, 4, 0)):
        ctx.set_hparams(sp_nms_radius=radius, sp_remove_borders=border, sp_topk_always=always)
        n, kxy, score, desc = ctx.extract(frames[:1], kmax=1024, thr=0.0005)
        r = oracle.superpoint(wd, frames[0], kmax=1024, thr=0.0005, nms_radius=radius, border=border, topk_always=bool(always))
        assert r["n"] > 0 and n[0] == r["n"] and np.array_equal(kxy[0], r["kxy"]) and np.array_equal(score[0], r["score"])
        assert np.array_equal(desc[0], r["desc"])
        below += int(always and r["n"] < 1024)
    assert below >= 1                                                 # the unconditional top-k ordered a set smaller than Kmax at least once
    with pytest.raises(capi.RfeError, match="sp_nms_radius"):
        ctx.set_hparams(sp_nms_radius=9)
    with pytest.raises(capi.RfeError, match="9 layers of 4 heads"):
        ctx.set_hparams(lg_layers=6)
    # a v2 LightGlue file of another depth is refused at load time, with the reason
    Wt.save(str(tmp_path / "lg6.rfew"), Wt.make_lightglue(seed=1), 2, {"layers": 6})
    with pytest.raises(capi.RfeError, match="hyper-parameters refused"):
        ctx.load_weights(lg_path=str(tmp_path / "lg6.rfew"))
    Wt.save(str(tmp_path / "lg.rfew"), Wt.make_lightglue(seed=1), 2, {"filter_threshold": 0.25})
    ctx.load_weights(lg_path=str(tmp_path / "lg.rfew"))
    assert ctx.get_hparams()["lg_filter_threshold"] == np.float32(0.25)
    ctx.close()
