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


def _node(op, inputs, outputs, name=""):
    body = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    return body + _ld(3, name.encode()) + _ld(4, op.encode())


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


def test_lightglue_roundtrip(tmp_path):
    blob = Wt.make_lightglue(seed=5)
    t = _named(blob, Wt.lg_manifest()[0])
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
    path = tmp_path / "lightglue_sim.onnx"
    path.write_bytes(_model(inits, nodes))
    got = OW.convert_lightglue(str(path))
    assert np.array_equal(got, blob)


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
