"""the dependency-free ONNX initializer reader and the mapping onto the canonical RFEW blobs.
The real superpoint.onnx / lightglue_sim.onnx are missing from the reference checkout, so the test writes
its own ONNX files (minimal protobuf writer below) using the public module naming the converter assumes, with
Linear layers as anonymous MatMul constants + named bias Adds -- the shape a torch.onnx export has."""
import struct

import numpy as np
import pytest

from rover_slam_amd import onnx_weights as OW, weights as Wt


# ---- minimal protobuf writer -------------------------------------------------------------------------------
def _vi(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _ld(fn, payload):
    return _vi((fn << 3) | 2) + _vi(len(payload)) + payload


def _tensor(name, arr):
    arr = np.ascontiguousarray(arr, np.float32)
    body = b"".join(_vi((1 << 3) | 0) + _vi(d) for d in arr.shape)      # dims
    body += _vi((2 << 3) | 0) + _vi(1)                                     # data_type = FLOAT
    body += _ld(8, name.encode()) + _ld(9, arr.tobytes())                  # name, raw_data
    return body


def _node(op, inputs, outputs, name="", attrs=None):
    body = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    body += _ld(3, name.encode()) + _ld(4, op.encode())
    for k, v in (attrs or {}).items():        # AttributeProto{name = 1, i = 3, type = 20 (INT = 2)}
        body += _ld(5, _ld(1, k.encode()) + _vi((3 << 3) | 0) + _vi(v) + _vi((20 << 3) | 0) + _vi(2))
    return body


def _model(initializers, nodes):
    graph = b"".join(_ld(1, n) for n in nodes) + b"".join(_ld(5, _tensor(k, v)) for k, v in initializers)
    return _vi((1 << 3) | 0) + _vi(8) + _ld(7, graph)                      # ir_version, graph


def _named(blob, manifest):
    return {name: blob[off:off + int(np.prod(shape))].reshape(shape) for name, off, shape in manifest}


# ---- SuperPoint ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("anonymous", [False, True])
def test_superpoint_roundtrip(tmp_path, anonymous):
    blob = Wt.make_superpoint(seed=3)
    t = _named(blob, Wt.sp_manifest()[0])
    inits = []
    for li, (name, cin, cout, k) in enumerate(Wt.SP_LAYERS):
        wn, bn = (f"onnx::Conv_{200 + 2 * li}", f"onnx::Conv_{201 + 2 * li}") if anonymous else (name + ".weight", name + ".bias")
        inits += [(wn, t[name + ".weight"]), (bn, t[name + ".bias"])]
    p = tmp_path / "superpoint.onnx"
    p.write_bytes(_model(inits, [_node("Conv", ["image", inits[0][0], inits[1][0]], ["x1"], "/conv1a/Conv")]))
    got = OW.convert_superpoint(str(p))
    assert np.array_equal(got, blob)


def test_superpoint_missing_tensor_is_reported(tmp_path):
    p = tmp_path / "bad.onnx"
    p.write_bytes(_model([("conv1a.weight", np.zeros((64, 1, 3, 3), np.float32))], []))
    with pytest.raises(ValueError, match="cannot place SuperPoint tensors"):
        OW.convert_superpoint(str(p))


# ---- LightGlue -----------------------------------------------------------------------------------------------
def _interleave_qkv(w, b):
    """canonical (t, h, d) rows -> the published module's (h, d, t) rows (inverse of the converter's step)."""
    return (np.ascontiguousarray(w.reshape(3, 4, 64, 256).transpose(1, 2, 0, 3).reshape(768, 256)),
            np.ascontiguousarray(b.reshape(3, 4, 64).transpose(1, 2, 0).reshape(768)))


def _named_lightglue_model(t):
    """cvg parameter names, every Linear as MatMul(x, anonymous W^T) + Add(named bias): what a constant-folded export looks like"""
    inits, nodes, cnt = [], [], [1000]

    def linear(prefix, w, b):     # MatMul(x, W^T as an anonymous constant) + Add(named bias)
        wname = f"onnx::MatMul_{cnt[0]}"; cnt[0] += 1
        inits.extend([(wname, np.ascontiguousarray(w.T)), (prefix + ".bias", b)])
        nodes.append(_node("MatMul", [f"{prefix}_in", wname], [f"{prefix}_mm"], f"/{prefix}/MatMul"))
        nodes.append(_node("Add", [prefix + ".bias", f"{prefix}_mm"], [f"{prefix}_out"], f"/{prefix}/Add"))

    inits.append(("onnx::MatMul_999", np.ascontiguousarray(t["posenc.Wr"].T)))   # bias-free Linear(2, 32)
    for l in range(Wt.LG_LAYERS):
        p, s, c = f"layers.{l}.", f"transformers.{l}.self_attn.", f"transformers.{l}.cross_attn."
        linear(s + "Wqkv", *_interleave_qkv(t[p + "self.Wqkv"], t[p + "self.bqkv"]))
        linear(s + "out_proj", t[p + "self.Wo"], t[p + "self.bo"])
        linear(s + "ffn.0", t[p + "self.W1"], t[p + "self.b1"])
        inits += [(s + "ffn.1.weight", t[p + "self.ln_g"]), (s + "ffn.1.bias", t[p + "self.ln_b"])]
        linear(s + "ffn.3", t[p + "self.W2"], t[p + "self.b2"])
        linear(c + "to_qk", t[p + "cross.Wqk"], t[p + "cross.bqk"])
        linear(c + "to_v", t[p + "cross.Wv"], t[p + "cross.bv"])
        linear(c + "to_out", t[p + "cross.Wo"], t[p + "cross.bo"])
        linear(c + "ffn.0", t[p + "cross.W1"], t[p + "cross.b1"])
        inits += [(c + "ffn.1.weight", t[p + "cross.ln_g"]), (c + "ffn.1.bias", t[p + "cross.ln_b"])]
        linear(c + "ffn.3", t[p + "cross.W2"], t[p + "cross.b2"])
    linear(f"log_assignment.{Wt.LG_LAYERS - 1}.final_proj", t["final_proj.W"], t["final_proj.b"])
    linear(f"log_assignment.{Wt.LG_LAYERS - 1}.matchability", t["matchability.w"].reshape(1, 256), t["matchability.b"])
    return _model(inits, nodes)


def test_lightglue_roundtrip(tmp_path):
    blob = Wt.make_lightglue(seed=5)
    path = tmp_path / "lightglue_sim.onnx"
    path.write_bytes(_named_lightglue_model(_named(blob, Wt.lg_manifest()[0])))
    got = OW.convert_lightglue(str(path))
    assert np.array_equal(got, blob)


def anonymous_cases(tmp_path):
    """every hand-written file of this module as (path, kind, converts?) -- tests/test_onnx_cpp.py runs the C++ reader over the same files"""
    out = []

    def put(name, data, kind, ok):
        p = tmp_path / name
        p.write_bytes(data)
        out.append((str(p), kind, ok))

    sp = _named(Wt.make_superpoint(seed=3), Wt.sp_manifest()[0])
    for anonymous in (False, True):
        inits = []
        for li, (name, cin, cout, k) in enumerate(Wt.SP_LAYERS):
            wn, bn = (f"onnx::Conv_{200 + 2 * li}", f"onnx::Conv_{201 + 2 * li}") if anonymous else (name + ".weight", name + ".bias")
            inits += [(wn, sp[name + ".weight"]), (bn, sp[name + ".bias"])]
        put(f"sp_{int(anonymous)}.onnx", _model(inits, [_node("Conv", ["image", inits[0][0], inits[1][0]], ["x1"], "/conv1a/Conv")]), 1, True)
    put("sp_bad.onnx", _model([("conv1a.weight", np.zeros((64, 1, 3, 3), np.float32))], []), 1, False)
    t = _named(Wt.make_lightglue(seed=6), Wt.lg_manifest()[0])
    put("lg_named.onnx", _named_lightglue_model(t), 2, True)
    put("lg_simplified.onnx", _simplified_lightglue(t), 2, True)
    put("lg_dropped.onnx", _simplified_lightglue(t, drop=(3, "cross.Wv")), 2, False)
    put("lg_extra.onnx", _simplified_lightglue(t, extra_first=True), 2, False)
    put("lg_bad.onnx", _model([("posenc.Wr.weight", np.zeros((32, 2), np.float32))], []), 2, False)
    return out


def _simplified_lightglue(t, drop=None, extra_first=False):
    """A file shaped like onnx-simplifier output: every initializer renamed to a numeric id, no node names, Linear layers
    as Gemm (alternating transB = 1 with [out,in] and transB = 0 with [in,out] constants), LayerNormalization nodes, the
    self block's weights used twice (image 0, image 1), nodes in execution order."""
    inits, nodes, cnt, flip = [], [], [0], [0]

    def const(arr):
        cnt[0] += 1
        inits.append((str(cnt[0]), np.ascontiguousarray(arr, np.float32)))
        return str(cnt[0])

    def gemm(w, b, uses=1):
        flip[0] ^= 1
        tb = flip[0]
        wn, bn = const(w if tb else w.T), const(b)
        for u in range(uses):
            nodes.append(_node("Gemm", [f"x{cnt[0]}_{u}", wn, bn], [f"y{cnt[0]}_{u}"], "", {"transB": tb} if tb else {"alpha_i": 1}))

    def ln(g, b, uses=1):
        gn, bn = const(g), const(b)
        for u in range(uses):
            nodes.append(_node("LayerNormalization", [f"h{cnt[0]}_{u}", gn, bn], [f"n{cnt[0]}_{u}"], "", {"axis": 1}))

    if extra_first:
        gemm(np.zeros((256, 256), np.float32), np.zeros(256, np.float32))
    wr = const(t["posenc.Wr"].T)
    nodes.append(_node("MatMul", ["kpts", wr], ["theta"]))
    for l in range(Wt.LG_LAYERS):
        p = f"layers.{l}."
        steps = [("self.Wqkv", "self.bqkv", 2), ("self.Wo", "self.bo", 2), ("self.W1", "self.b1", 2), "ln_self", ("self.W2", "self.b2", 2),
                 ("cross.Wqk", "cross.bqk", 2), ("cross.Wv", "cross.bv", 2), ("cross.Wo", "cross.bo", 2), ("cross.W1", "cross.b1", 2), "ln_cross",
                 ("cross.W2", "cross.b2", 2)]
        for st in steps:
            if st == "ln_self":
                ln(t[p + "self.ln_g"], t[p + "self.ln_b"], 2)
            elif st == "ln_cross":
                ln(t[p + "cross.ln_g"], t[p + "cross.ln_b"], 2)
            elif (l, st[0]) != drop:
                w, b = t[p + st[0]], t[p + st[1]]
                if st[0] == "self.Wqkv":
                    w, b = _interleave_qkv(w, b)
                gemm(w, b, st[2])
    gemm(t["final_proj.W"], t["final_proj.b"], 2)
    gemm(t["matchability.w"].reshape(1, 256), t["matchability.b"], 2)
    return _model(inits, nodes)


def test_lightglue_simplifier_style_file(tmp_path):
    """initializers renamed to numeric ids, Gemm instead of MatMul + Add, transposed weights."""
    blob = Wt.make_lightglue(seed=6)
    t = _named(blob, Wt.lg_manifest()[0])
    path = tmp_path / "lightglue_sim.onnx"
    path.write_bytes(_simplified_lightglue(t))
    inits, nodes = OW.read_model(str(path))
    assert all(k.isdigit() for k in inits) and {n["op"] for n in nodes} == {"Gemm", "LayerNormalization", "MatMul"}
    assert any(n["attrs"].get("transB") == 1 for n in nodes) and any("transB" not in n["attrs"] for n in nodes if n["op"] == "Gemm")
    assert np.array_equal(OW.convert_lightglue(str(path)), blob)
    # a file that deviates is not guessed at: the first Linear that does not fit is named with position and shapes
    path.write_bytes(_simplified_lightglue(t, drop=(3, "cross.Wv")))
    with pytest.raises(ValueError) as e:
        OW.convert_lightglue(str(path))
    msg = str(e.value)
    assert "83 Linear layers in the graph" in msg and "has 84" in msg and "expected layers.3.cross.Wo (256, 256) + bias, found (512, 512)" in msg
    path.write_bytes(_simplified_lightglue(t, extra_first=True))     # e.g. an input projection the published graph does not have
    with pytest.raises(ValueError, match=r"Linear #0: expected posenc.Wr \(32, 2\), found \(256, 256\)"):
        OW.convert_lightglue(str(path))


def test_lightglue_incomplete_file_is_reported(tmp_path):
    p = tmp_path / "bad.onnx"
    p.write_bytes(_model([("posenc.Wr.weight", np.zeros((32, 2), np.float32))], []))
    with pytest.raises(ValueError, match="cannot place LightGlue tensors"):
        OW.convert_lightglue(str(p))


def test_ort_parity_tool_reports_missing_prerequisites():
    """tools/ort_parity.py is a guarded harness: without onnxruntime + the blobs it says so (rc 2)."""
    import importlib.util, os, subprocess, sys
    if importlib.util.find_spec("onnxruntime") is not None:
        import pytest
        pytest.skip("onnxruntime present: the tool would run for real")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "ort_parity.py"), "--superpoint", "missing.onnx"],
                       capture_output=True, text=True)
    assert r.returncode == 2 and "unpinned" in r.stderr
