"""Real-weight ingestion: read the initializers of the reference's two model files
(`onnxmodel/superpoint.onnx`, `onnxmodel/lightglue_sim.onnx`; src/Extractors/SPextractor.cc:93,
src/Matchers/lightglue_onnx.cpp:38) without onnx / protobuf installed, and re-pack them into the canonical
RFEW blobs of `weights.py`.

UNVALIDATED AGAINST THE REAL BLOBS: both files are missing from the reference checkout
(.MISSING_LARGE_BLOBS:4-5).  The tensor naming assumed here is that of the public SuperPoint / LightGlue
PyTorch modules the LightGlue-ONNX style exports are traced from (`conv1a.weight` ..., `transformers.{i}.
self_attn.Wqkv` ...); Linear weights that the exporter turned into anonymous MatMul constants are recovered
through the graph (MatMul -> Add(bias with the parameter's name), or Gemm).  Files that went through onnx-simplifier
keep no parameter names at all (initializers renamed to numeric ids, MatMul + Add fused into Gemm, weights stored
[in,out] or [out,in] according to transB): for those the Linear layers are taken in order of FIRST USE in the
(topologically sorted) node list and checked, one by one, against the shape sequence of the published LightGlue graph;
the first disagreement is reported with its position, the expected and the found shape -- nothing is guessed past it.
`tests/test_onnx_weights.py` exercises reader and both mappings on ONNX files written by the test itself; when a real
file deviates, `convert_*` raises and lists what it could not place.

Only a protobuf *wire-format* reader is implemented (ModelProto.graph.{initializer,node}); external-data
tensors are not supported.
"""
import struct

import numpy as np

from . import weights as Wt


# ---------------------------------------------------------------- protobuf wire format
def _varint(buf, i):
    r, shift = 0, 0
    while True:
        b = buf[i]
        i += 1
        r |= (b & 0x7F) << shift
        if not b & 0x80:
            return r, i
        shift += 7


def _fields(buf):
    """yield (field_number, wire_type, value) for one message; length-delimited values are memoryviews."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v = bytes(buf[i:i + 8]); i += 8
        elif wt == 2:
            ln, i = _varint(buf, i)
            v = buf[i:i + ln]; i += ln
        elif wt == 5:
            v = bytes(buf[i:i + 4]); i += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fn, wt, v


_DT = {1: np.float32, 7: np.int64, 6: np.int32, 11: np.float64, 10: np.float16}


def _tensor(buf):
    dims, dtype, name, raw, floats, int64s = [], 1, "", None, [], []
    for fn, wt, v in _fields(buf):
        if fn == 1:      # dims
            if wt == 2:
                j, vv = 0, bytes(v)
                while j < len(vv):
                    d, j = _varint(vv, j)
                    dims.append(d)
            else:
                dims.append(v)
        elif fn == 2:
            dtype = v
        elif fn == 8:
            name = bytes(v).decode()
        elif fn == 9:
            raw = bytes(v)
        elif fn == 4:    # float_data
            floats.append(np.frombuffer(bytes(v), "<f4") if wt == 2 else np.frombuffer(v, "<f4"))
        elif fn == 7:    # int64_data
            if wt == 2:
                j, vv = 0, bytes(v)
                while j < len(vv):
                    d, j = _varint(vv, j)
                    int64s.append(d)
            else:
                int64s.append(v)
        elif fn == 13:
            raise ValueError(f"tensor {name!r}: external data is not supported")
    if dtype not in _DT:
        return name, None
    if raw is not None:
        arr = np.frombuffer(raw, np.dtype(_DT[dtype]).newbyteorder("<")).astype(_DT[dtype])
    elif floats:
        arr = np.concatenate(floats).astype(np.float32)
    elif int64s:
        arr = np.array(int64s, np.int64)
    else:
        arr = np.zeros(0, _DT[dtype])
    return name, arr.reshape(dims) if dims else arr


def read_model(path):
    """Returns (initializers: name -> ndarray, nodes: list of dict(op, inputs, outputs, name))."""
    data = memoryview(open(path, "rb").read())
    inits, nodes = {}, []
    for fn, wt, v in _fields(data):
        if fn != 7 or wt != 2:        # ModelProto.graph
            continue
        for gfn, gwt, gv in _fields(v):
            if gfn == 5 and gwt == 2:  # GraphProto.initializer
                name, arr = _tensor(gv)
                if arr is not None:
                    inits[name] = arr
            elif gfn == 1 and gwt == 2:  # GraphProto.node
                node = dict(op="", inputs=[], outputs=[], name="", attrs={})
                for nfn, nwt, nv in _fields(gv):
                    if nfn == 1:
                        node["inputs"].append(bytes(nv).decode())
                    elif nfn == 2:
                        node["outputs"].append(bytes(nv).decode())
                    elif nfn == 3:
                        node["name"] = bytes(nv).decode()
                    elif nfn == 4:
                        node["op"] = bytes(nv).decode()
                    elif nfn == 5 and nwt == 2:   # AttributeProto: integer attributes only (transB, axis ...)
                        aname, aval = "", None
                        for afn, awt, av in _fields(nv):
                            if afn == 1:
                                aname = bytes(av).decode()
                            elif afn == 3 and awt == 0:
                                aval = av
                        if aval is not None:
                            node["attrs"][aname] = aval
                nodes.append(node)
    return inits, nodes


def _linears_from_graph(inits, nodes):
    """Recover `prefix -> (W [out,in], b)` for Linear layers exported as MatMul(x, W^T) + Add(bias) or Gemm.
    The bias initializer keeps the PyTorch parameter name (`<prefix>.bias`)."""
    producer = {o: n for n in nodes for o in n["outputs"]}
    out = {}
    for n in nodes:
        if n["op"] == "Add":
            bias = [i for i in n["inputs"] if i in inits and i.endswith(".bias")]
            other = [i for i in n["inputs"] if i not in inits]
            if len(bias) == 1 and len(other) == 1 and other[0] in producer and producer[other[0]]["op"] == "MatMul":
                mm = producer[other[0]]
                w = [i for i in mm["inputs"] if i in inits]
                if len(w) == 1 and inits[w[0]].ndim == 2:
                    out[bias[0][:-5]] = (np.ascontiguousarray(inits[w[0]].T), inits[bias[0]])
        elif n["op"] == "Gemm" and len(n["inputs"]) >= 3 and n["inputs"][2] in inits and n["inputs"][2].endswith(".bias"):
            w, b = inits.get(n["inputs"][1]), inits[n["inputs"][2]]
            if w is not None and w.ndim == 2:
                # Gemm keeps PyTorch's [out,in] with transB=1; if the exporter stored [in,out] the shapes tell
                out[n["inputs"][2][:-5]] = (w if w.shape[0] == b.shape[0] else np.ascontiguousarray(w.T), b)
    return out


def _linears_in_order_of_use(inits, nodes):
    """Name-free view of the graph: every Linear parameter set, once, in order of first use in the node list, as
    (W [out,in], bias or None, where).  Gemm: B is [out,in] when transB = 1, [in,out] otherwise (ONNX semantics);
    MatMul: the constant operand is W^T; its bias is the 1-D constant of the Add that consumes the product."""
    consumers = {}
    for n in nodes:
        for i in n["inputs"]:
            consumers.setdefault(i, []).append(n)
    seen, out = set(), []
    for idx, n in enumerate(nodes):
        if n["op"] == "Gemm" and len(n["inputs"]) >= 2 and n["inputs"][1] in inits and inits[n["inputs"][1]].ndim == 2:
            wn = n["inputs"][1]
            if wn in seen:
                continue
            seen.add(wn)
            w = inits[wn] if n["attrs"].get("transB", 0) == 1 else inits[wn].T
            b = inits.get(n["inputs"][2]) if len(n["inputs"]) >= 3 else None
            out.append((np.ascontiguousarray(w), b, f"node {idx} Gemm({wn})"))
        elif n["op"] == "MatMul":
            wn = [i for i in n["inputs"] if i in inits and inits[i].ndim == 2]
            if len(wn) != 1 or wn[0] in seen:
                continue
            seen.add(wn[0])
            b = None
            for c in consumers.get(n["outputs"][0] if n["outputs"] else "", []):
                if c["op"] == "Add":
                    cb = [i for i in c["inputs"] if i in inits and inits[i].ndim == 1]
                    if len(cb) == 1:
                        b = inits[cb[0]]
            out.append((np.ascontiguousarray(inits[wn[0]].T), b, f"node {idx} MatMul({wn[0]})"))
    return out


def _layernorms_in_order_of_use(inits, nodes, width=512):
    """(gamma, beta) of every LayerNorm(width), once, in order of first use: LayerNormalization nodes, or the decomposed
    form's Mul(x_hat, gamma) -> Add(beta) with two 1-D constants of that width."""
    consumers = {}
    for n in nodes:
        for i in n["inputs"]:
            consumers.setdefault(i, []).append(n)
    seen, out = set(), []
    for n in nodes:
        if n["op"] == "LayerNormalization" and len(n["inputs"]) >= 3 and n["inputs"][1] in inits and inits[n["inputs"][1]].shape == (width,):
            if n["inputs"][1] not in seen:
                seen.add(n["inputs"][1])
                out.append((inits[n["inputs"][1]], inits[n["inputs"][2]]))
        elif n["op"] == "Mul":
            g = [i for i in n["inputs"] if i in inits and inits[i].shape == (width,)]
            if len(g) == 1 and g[0] not in seen:
                for c in consumers.get(n["outputs"][0] if n["outputs"] else "", []):
                    bb = [i for i in c["inputs"] if c["op"] == "Add" and i in inits and inits[i].shape == (width,)]
                    if len(bb) == 1:
                        seen.add(g[0])
                        out.append((inits[g[0]], inits[bb[0]]))
                        break
    return out


# ---------------------------------------------------------------- SuperPoint
def convert_superpoint(path):
    """-> float32 blob in the canonical SuperPoint layout (weights.sp_manifest)."""
    inits, _ = read_model(path)
    man, n = Wt.sp_manifest()
    blob = np.empty(n, np.float32)
    # fallback when parameter names were not preserved: conv weights / biases in file order
    convs = [a for a in inits.values() if a.ndim == 4]
    biases = [a for a in inits.values() if a.ndim == 1 and a.dtype == np.float32]
    missing = []
    for li, (name, cin, cout, k) in enumerate(Wt.SP_LAYERS):
        for leaf, shape, pool in (("weight", (cout, cin, k, k), convs), ("bias", (cout,), biases)):
            key = f"{name}.{leaf}"
            arr = inits.get(key)
            if arr is None:
                cand = [a for a in pool if a.shape == shape]
                # layers sharing a shape (conv1b, conv2a, conv2b ...) are told apart by their order in the file
                same_before = sum(1 for (nm, ci, co, kk) in Wt.SP_LAYERS[:li]
                                  if ((co, ci, kk, kk) if leaf == "weight" else (co,)) == shape)
                arr = cand[same_before] if same_before < len(cand) else None
            if arr is None or tuple(arr.shape) != shape:
                missing.append(key)
                continue
            off = next(o for nm, o, _ in man if nm == key)
            blob[off:off + arr.size] = arr.astype(np.float32).ravel()
    if missing:
        raise ValueError(f"{path}: cannot place SuperPoint tensors {missing}; initializers present: {sorted(inits)[:40]}")
    return blob


# ---------------------------------------------------------------- LightGlue
def _deinterleave_qkv(w, b):
    """published LightGlue SelfBlock: qkv = Wqkv(x).unflatten(-1, (heads, -1, 3)) -> output row h*192 + d*3 + t;
    canonical layout here: row t*256 + h*64 + d."""
    w4 = w.reshape(4, 64, 3, 256).transpose(2, 0, 1, 3).reshape(768, 256)
    b4 = b.reshape(4, 64, 3).transpose(2, 0, 1).reshape(768)
    return np.ascontiguousarray(w4), np.ascontiguousarray(b4)


def convert_lightglue(path, n_layers=Wt.LG_LAYERS):
    """-> float32 blob in the canonical LightGlue layout (weights.lg_manifest)."""
    inits, nodes = read_model(path)
    lin = _linears_from_graph(inits, nodes)

    def linear(prefix):
        if prefix + ".weight" in inits and prefix + ".bias" in inits:
            return inits[prefix + ".weight"], inits[prefix + ".bias"]
        if prefix in lin:
            return lin[prefix]
        raise KeyError(prefix)

    t, missing = {}, []

    def put(name, arr):
        t[name] = np.asarray(arr, np.float32)

    named = any(k.startswith("transformers.") for k in inits)
    if not named:
        return _convert_lightglue_by_structure(path, inits, nodes, n_layers)
    try:
        wr = inits.get("posenc.Wr.weight")
        if wr is None:   # bias-free Linear: anonymous MatMul constant [2,32]
            cand = [a for a in inits.values() if a.ndim == 2 and a.shape in ((2, 32), (32, 2))]
            wr = cand[0] if cand else None
        put("posenc.Wr", wr if wr.shape == (32, 2) else wr.T)
    except Exception:
        missing.append("posenc.Wr")
    for l in range(n_layers):
        p, s, c = f"layers.{l}.", f"transformers.{l}.self_attn.", f"transformers.{l}.cross_attn."
        try:
            w, b = _deinterleave_qkv(*linear(s + "Wqkv"))
            put(p + "self.Wqkv", w); put(p + "self.bqkv", b)
            for src, dst in ((s + "out_proj", "self.Wo:self.bo"), (s + "ffn.0", "self.W1:self.b1"), (s + "ffn.3", "self.W2:self.b2"),
                             (c + "to_qk", "cross.Wqk:cross.bqk"), (c + "to_v", "cross.Wv:cross.bv"), (c + "to_out", "cross.Wo:cross.bo"),
                             (c + "ffn.0", "cross.W1:cross.b1"), (c + "ffn.3", "cross.W2:cross.b2")):
                w, b = linear(src)
                wn, bn = dst.split(":")
                put(p + wn, w); put(p + bn, b)
            for src, tag in ((s + "ffn.1", "self"), (c + "ffn.1", "cross")):   # LayerNorm(512)
                put(p + tag + ".ln_g", inits[src + ".weight"]); put(p + tag + ".ln_b", inits[src + ".bias"])
        except KeyError as e:
            missing.append(str(e))
    try:   # no early exit in the fused export: only the last assignment head is live
        w, b = linear(f"log_assignment.{n_layers - 1}.final_proj")
        put("final_proj.W", w); put("final_proj.b", b)
        w, b = linear(f"log_assignment.{n_layers - 1}.matchability")
        put("matchability.w", w.reshape(256)); put("matchability.b", b.reshape(1))
    except KeyError as e:
        missing.append(str(e))
    man, n = Wt.lg_manifest()
    blob = np.empty(n, np.float32)
    for name, off, shape in man:
        arr = t.get(name)
        if arr is None or tuple(arr.shape) != tuple(shape):
            missing.append(f"{name}{'' if arr is None else ' shape ' + str(arr.shape)}")
            continue
        blob[off:off + arr.size] = arr.ravel()
    if missing:
        raise ValueError(f"{path}: cannot place LightGlue tensors {missing[:12]}...; linears recovered: {sorted(lin)[:20]}")
    return blob


def _convert_lightglue_by_structure(path, inits, nodes, n_layers):
    """No parameter names survive (onnx-simplifier output): Linear layers in order of first use against the shape sequence
    of the published graph -- posenc (32x2, no bias); per layer self {Wqkv 768x256, out_proj 256x256, ffn.0 512x512,
    ffn.3 256x512} then cross {to_qk, to_v, to_out 256x256, ffn.0 512x512, ffn.3 256x512}; final_proj 256x256,
    matchability 1x256 -- and LayerNorm(512) pairs in order of first use (self, cross per layer)."""
    lins = _linears_in_order_of_use(inits, nodes)
    lns = _layernorms_in_order_of_use(inits, nodes)
    expect = [("posenc.Wr", None, (32, 2), False)]
    for l in range(n_layers):
        p = f"layers.{l}."
        expect += [(p + "self.Wqkv", p + "self.bqkv", (768, 256), True), (p + "self.Wo", p + "self.bo", (256, 256), True),
                   (p + "self.W1", p + "self.b1", (512, 512), True), (p + "self.W2", p + "self.b2", (256, 512), True),
                   (p + "cross.Wqk", p + "cross.bqk", (256, 256), True), (p + "cross.Wv", p + "cross.bv", (256, 256), True),
                   (p + "cross.Wo", p + "cross.bo", (256, 256), True), (p + "cross.W1", p + "cross.b1", (512, 512), True),
                   (p + "cross.W2", p + "cross.b2", (256, 512), True)]
    expect += [("final_proj.W", "final_proj.b", (256, 256), True), ("matchability.w", "matchability.b", (1, 256), True)]
    problems = []
    if len(lins) != len(expect):
        problems.append(f"{len(lins)} Linear layers in the graph, the published {n_layers}-layer LightGlue has {len(expect)}")
    t = {}
    for pos, (wname, bname, shape, has_bias) in enumerate(expect):
        if pos >= len(lins):
            problems.append(f"Linear #{pos} ({wname} {shape}): graph has no more Linear layers")
            break
        w, b, where = lins[pos]
        if tuple(w.shape) != shape or (has_bias and (b is None or b.shape != (shape[0],))):
            problems.append(f"Linear #{pos}: expected {wname} {shape}{' + bias' if has_bias else ''}, found {tuple(w.shape)}"
                            f"{'' if b is None else ' + bias ' + str(tuple(b.shape))} at {where}")
            break          # everything after a disagreement would be guesswork
        if wname.endswith("self.Wqkv"):
            w, b = _deinterleave_qkv(w, b)
        t[wname] = w.reshape(256) if wname == "matchability.w" else w
        if bname:
            t[bname] = b.reshape(-1)
    if len(lns) != 2 * n_layers:
        problems.append(f"{len(lns)} LayerNorm(512) parameter pairs found, expected {2 * n_layers}")
    else:
        for l in range(n_layers):
            for j, tag in enumerate(("self", "cross")):
                t[f"layers.{l}.{tag}.ln_g"], t[f"layers.{l}.{tag}.ln_b"] = lns[2 * l + j]
    man, n = Wt.lg_manifest()
    blob = np.empty(n, np.float32)
    if not problems:
        for name, off, shape in man:
            arr = t.get(name)
            if arr is None or tuple(arr.shape) != tuple(shape):
                problems.append(f"{name}: not placed")
                continue
            blob[off:off + arr.size] = np.asarray(arr, np.float32).ravel()
    if problems:
        raise ValueError(f"{path}: cannot place LightGlue tensors by structure (no parameter names in the file): " + "; ".join(problems[:6]))
    return blob


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="convert superpoint.onnx / lightglue_sim.onnx initializers to RFEW containers")
    ap.add_argument("--superpoint"); ap.add_argument("--lightglue"); ap.add_argument("--out-dir", default=".")
    a = ap.parse_args(argv)
    import os
    if a.superpoint:
        Wt.save(os.path.join(a.out_dir, "superpoint.rfew"), convert_superpoint(a.superpoint), 1)
    if a.lightglue:
        Wt.save(os.path.join(a.out_dir, "lightglue_sim.rfew"), convert_lightglue(a.lightglue), 2)


if __name__ == "__main__":
    main()
