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

Besides the weights, the graph's baked-in hyper-parameters (max_num_keypoints, detection threshold, NMS radius, border; LightGlue
depth, heads, filter threshold) are read from its nodes and written into the RFEW v2 header (`read_*_hparams`, `convert`, `main`).

Only a protobuf *wire-format* reader is implemented (ModelProto.graph.{initializer,node} with node attributes); external-data
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


_DT = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64,
       12: np.uint32, 13: np.uint64}


def _tensor(buf):
    dims, dtype, name, raw, floats, int64s, int32s, doubles = [], 1, "", None, [], [], [], []
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
                    int64s.append(d - (1 << 64) if d >= (1 << 63) else d)
            else:
                int64s.append(v - (1 << 64) if v >= (1 << 63) else v)
        elif fn == 5:    # int32_data (also carries bool / int8 / int16 / uint8 / float16 bit patterns)
            if wt == 2:
                j, vv = 0, bytes(v)
                while j < len(vv):
                    d, j = _varint(vv, j)
                    int32s.append(d - (1 << 64) if d >= (1 << 63) else d)
            else:
                int32s.append(v - (1 << 64) if v >= (1 << 63) else v)
        elif fn == 10:   # double_data
            doubles.append(np.frombuffer(bytes(v), "<f8"))
        elif fn == 13:
            raise ValueError(f"tensor {name!r}: external data is not supported")
    if dtype not in _DT:
        return name, None
    if raw is not None:
        arr = np.frombuffer(raw, np.dtype(_DT[dtype]).newbyteorder("<") if dtype != 9 else np.uint8).astype(_DT[dtype])
    elif floats:
        arr = np.concatenate(floats).astype(np.float32)
    elif int64s:
        arr = np.array(int64s, np.int64)
    elif int32s and dtype != 10:
        arr = np.array(int32s, np.int64).astype(_DT[dtype])
    elif doubles:
        arr = np.concatenate(doubles).astype(np.float64)
    else:
        arr = np.zeros(0, _DT[dtype])
    if not dims:         # no dims: a rank-0 tensor when it holds one element (Constant scalars), an empty one otherwise
        return name, arr.reshape(()) if arr.size == 1 else arr
    return name, arr.reshape(dims)


def _attribute(buf):
    """AttributeProto -> (name, value): i (int), f (float), s (str), t (ndarray), ints / floats (list).  Graph-valued attributes
    (If / Loop bodies) come back as the marker string "<graph>": their presence is what matters here, not their content."""
    name, val, ints, floats = "", None, [], []
    for fn, wt, v in _fields(buf):
        if fn == 1:
            name = bytes(v).decode()
        elif fn == 2 and wt == 5:
            val = struct.unpack("<f", v)[0]
        elif fn == 3 and wt == 0:
            val = v - (1 << 64) if v >= (1 << 63) else v
        elif fn == 4 and wt == 2:
            val = bytes(v).decode(errors="replace")
        elif fn == 5 and wt == 2:
            val = _tensor(v)[1]
        elif fn == 6 and wt == 2:
            val = "<graph>"
        elif fn == 7:     # floats
            floats.extend(np.frombuffer(bytes(v), "<f4").tolist() if wt == 2 else [struct.unpack("<f", v)[0]])
        elif fn == 8:     # ints
            if wt == 2:
                j, vv = 0, bytes(v)
                while j < len(vv):
                    d, j = _varint(vv, j)
                    ints.append(d - (1 << 64) if d >= (1 << 63) else d)
            else:
                ints.append(v - (1 << 64) if v >= (1 << 63) else v)
    if val is None and ints:
        val = ints
    if val is None and floats:
        val = floats
    return name, val


def read_model(path):
    """Returns (initializers: name -> ndarray, nodes: list of dict(op, inputs, outputs, name, attrs))."""
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
                    elif nfn == 5 and nwt == 2:   # AttributeProto
                        aname, aval = _attribute(nv)
                        if aval is not None:
                            node["attrs"][aname] = aval
                nodes.append(node)
    return inits, nodes


def read_graph_io(path):
    """-> dict(inputs=[names the caller feeds, initializers excluded], outputs=[names], opset=int or None): the tensor names the
    reference binds by string (src/Extractors/superpoint_onnx.cc:100,133-134; src/Matchers/lightglue_onnx.cpp:168-172,210-211)."""
    data = memoryview(open(path, "rb").read())
    ins, outs, init_names, opset = [], [], set(), None
    for fn, wt, v in _fields(data):
        if fn == 8 and wt == 2:         # ModelProto.opset_import: OperatorSetIdProto{domain = 1, version = 2}
            dom, ver = "", None
            for ofn, owt, ov in _fields(v):
                if ofn == 1:
                    dom = bytes(ov).decode()
                elif ofn == 2:
                    ver = ov
            if dom in ("", "ai.onnx") and ver is not None:
                opset = int(ver)
        if fn != 7 or wt != 2:
            continue
        for gfn, gwt, gv in _fields(v):
            if gfn in (11, 12) and gwt == 2:     # GraphProto.input / output: ValueInfoProto{name = 1}
                for vfn, _, vv in _fields(gv):
                    if vfn == 1:
                        (ins if gfn == 11 else outs).append(bytes(vv).decode())
            elif gfn == 5 and gwt == 2:
                for tfn, _, tv in _fields(gv):
                    if tfn == 8:
                        init_names.add(bytes(tv).decode())
    return dict(inputs=[n for n in ins if n not in init_names], outputs=outs, opset=opset)


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


# ---------------------------------------------------------------- graph hyper-parameters
# The reference's C++ holds NONE of these: K is the shape of the `keypoints` output (src/Extractors/superpoint_onnx.cc:169-181) and
# matches0 / mscores0 arrive already filtered (src/Matchers/lightglue_onnx.cpp:404-409) because max_num_keypoints, the detection
# threshold, the NMS radius, the border width, LightGlue's depth / heads / filter threshold are constants of the two graphs.  The
# readers below recover them from the nodes a PyTorch export of the published modules produces (exercised on such exports in
# tests/test_onnx_hparams.py; the real files are missing, .MISSING_LARGE_BLOBS:4-5).  Nothing is guessed: what cannot be read is
# listed in `problems`, and convert / main refuse to write a weight file until the caller states the value explicitly (`assume`).
_PASS_THROUGH = ("Cast", "Reshape", "Unsqueeze", "Squeeze", "Identity", "Flatten")


def _graph_view(inits, nodes):
    """constants (initializers + Constant nodes) and the producer of every tensor name"""
    consts = dict(inits)
    for n in nodes:
        if n["op"] == "Constant" and n["outputs"]:
            a = n["attrs"]
            for key in ("value", "value_float", "value_int", "value_floats", "value_ints"):
                if key in a and not isinstance(a[key], str):
                    consts[n["outputs"][0]] = np.asarray(a[key])
                    break
    producer = {o: n for n in nodes for o in n["outputs"] if o}
    return consts, producer


def _const_of(name, consts, producer, depth=6):
    """the constant a tensor name resolves to through shape-only pass-through ops, or None"""
    while depth >= 0:
        if name in consts:
            return consts[name]
        n = producer.get(name)
        if n is None or n["op"] not in _PASS_THROUGH or not n["inputs"]:
            return None
        name, depth = n["inputs"][0], depth - 1
    return None


def _scalar(v):
    if v is None:
        return None
    v = np.asarray(v)
    return v.reshape(-1)[0].item() if v.size == 1 else None


def _ancestors(name, producer, limit=400):
    """nodes upstream of a tensor name (breadth first, bounded)"""
    seen, out, todo = set(), [], [name]
    while todo and len(out) < limit:
        t = todo.pop(0)
        n = producer.get(t)
        if n is None or id(n) in seen:
            continue
        seen.add(id(n))
        out.append(n)
        todo.extend(i for i in n["inputs"] if i)
    return out


def read_superpoint_hparams(path):
    """-> (hparams dict over weights.SP_HPARAMS keys, values None where unreadable; problems: list of str).
    max_keypoints <- the k of TopK (through Min(k, n_candidates) / shape ops; a Min there means the top-k runs unconditionally:
    topk_always = 1); detection_threshold <- the one positive constant a
    Greater compares against (the suppression masks compare against 0); nms_radius <- kernel_shape / pads of the stride-1 MaxPools
    (the published simple_nms has 1 + 2 x 2 of them); remove_borders <- the constant Slice bounds ([:b] / [-b:]) of the index
    vectors the border ScatterNDs write -1 through; GridSample must be bilinear with align_corners = 1, zero padding."""
    inits, nodes = read_model(path)
    consts, producer = _graph_view(inits, nodes)
    hp = {k: None for k in Wt.SP_HPARAMS}
    problems = []
    # ---- NMS radius
    nms = []
    for n in nodes:
        if n["op"] != "MaxPool":
            continue
        ks, st = n["attrs"].get("kernel_shape"), n["attrs"].get("strides") or [1, 1]
        if isinstance(ks, list) and len(ks) == 2 and list(st) == [1, 1]:
            nms.append((tuple(ks), tuple(n["attrs"].get("pads") or [0, 0, 0, 0])))
    if not nms:
        problems.append("nms_radius: no stride-1 MaxPool in the graph (NMS not exported as max_pool2d?)")
    else:
        kinds = set(nms)
        (kh, kw), pads = nms[0]
        r = (kh - 1) // 2
        if len(kinds) != 1 or kh != kw or kh % 2 == 0 or set(pads) != {r}:
            problems.append(f"nms_radius: the stride-1 MaxPools disagree or are not (2r+1) windows padded by r: {sorted(kinds)}")
        elif len(nms) != 5:
            problems.append(f"nms_radius: {len(nms)} stride-1 MaxPools, the published simple_nms (2 suppression rounds) has 5 -- "
                            "the kernels implement exactly that recurrence")
        else:
            hp["nms_radius"] = r
    # ---- detection threshold
    thr = set()
    for n in nodes:
        if n["op"] == "Greater" and len(n["inputs"]) == 2:
            v = _scalar(_const_of(n["inputs"][1], consts, producer))
            if isinstance(v, float) and 0.0 < v < 1.0:
                thr.add(float(np.float32(v)))
    if len(thr) == 1:
        hp["detection_threshold"] = thr.pop()
    else:
        problems.append(f"detection_threshold: expected exactly one Greater(x, c) with 0 < c < 1, found constants {sorted(thr)}")
    # ---- max_num_keypoints
    topk = [n for n in nodes if n["op"] == "TopK"]
    if len(topk) != 1:
        problems.append(f"max_keypoints: {len(topk)} TopK nodes (an export without max_num_keypoints returns every candidate; "
                        "the C ABI needs a capacity: state it with assume)")
    else:
        t = topk[0]
        if t["attrs"].get("largest", 1) != 1:
            problems.append("max_keypoints: TopK with largest = 0")
        name, k = t["inputs"][1] if len(t["inputs"]) > 1 else "", None
        for _ in range(8):                      # K input: constant, or Min(constant, dynamic count) behind shape ops
            c = _scalar(_const_of(name, consts, producer))
            if c is not None:
                k = int(c)
                break
            n = producer.get(name)
            if n is None:
                break
            if n["op"] == "Min":                # torch.topk(scores, min(k, n)): applied whatever the candidate count -> always sorted
                cs = [_scalar(_const_of(i, consts, producer)) for i in n["inputs"]]
                cs = [int(c) for c in cs if c is not None]
                k = cs[0] if len(cs) == 1 else None
                hp["topk_always"] = 1
                break
            if n["op"] in _PASS_THROUGH and n["inputs"]:
                name = n["inputs"][0]
            else:
                break
        if k is None or k < 1:
            problems.append("max_keypoints: the K input of TopK does not resolve to a constant (or Min(constant, count))")
        else:
            hp["max_keypoints"] = k
            if hp["topk_always"] is None:       # constant k (guarded by control flow upstream): the published top_k_keypoints
                hp["topk_always"] = 0
    # ---- border
    pads, nscatter = set(), 0
    for n in nodes:
        if n["op"] != "ScatterND" or len(n["inputs"]) < 3:
            continue
        upd = _const_of(n["inputs"][2], consts, producer)
        anc = _ancestors(n["inputs"][1], producer)
        here = set()
        for a in anc:
            if a["op"] != "Slice" or len(a["inputs"]) < 3:
                continue
            src = producer.get(a["inputs"][0])
            while src is not None and src["op"] in _PASS_THROUGH and src["inputs"]:
                src = producer.get(src["inputs"][0])
            if src is None or src["op"] != "Range":
                continue                       # only slices of an index vector (arange(H) / arange(W)) describe the border
            st, en = _scalar(_const_of(a["inputs"][1], consts, producer)), _scalar(_const_of(a["inputs"][2], consts, producer))
            if st == 0 and en is not None and 0 < en <= 64:
                here.add(int(en))
            elif st is not None and -64 <= st < 0:
                here.add(int(-st))
        if here:
            nscatter += 1
            pads |= here
            if upd is not None and np.asarray(upd).size and not np.all(np.asarray(upd) == -1):
                problems.append("remove_borders: a border ScatterND writes something other than -1")
    if nscatter == 0:
        problems.append("remove_borders: no ScatterND over sliced index vectors found (border handled in another form, or not at all: "
                        "state it with assume, 0 = no border)")
    elif len(pads) != 1:
        problems.append(f"remove_borders: the border slices disagree: {sorted(pads)}")
    else:
        hp["remove_borders"] = pads.pop()
    # ---- descriptor sampling
    gs = [n for n in nodes if n["op"] == "GridSample"]
    if len(gs) != 1:
        problems.append(f"grid_sample: {len(gs)} GridSample nodes (descriptor sampling exported in another form)")
    else:
        a = gs[0]["attrs"]
        if a.get("mode", "bilinear") not in ("bilinear", "linear") or a.get("align_corners", 0) != 1 or a.get("padding_mode", "zeros") != "zeros":
            problems.append(f"grid_sample: mode / align_corners / padding_mode = {a.get('mode', 'bilinear')} / {a.get('align_corners', 0)} / "
                            f"{a.get('padding_mode', 'zeros')}; the kernels implement bilinear / 1 / zeros")
    return hp, problems


def read_lightglue_hparams(path):
    """-> (hparams dict over weights.LG_HPARAMS keys, problems).  layers <- the `transformers.{i}.` parameter names or the number of
    768 x 256 Linears; heads <- the head-split Reshape constants ([.., heads, 64(, 3)]); filter_threshold <- the one Greater(x, c)
    with 0 < c < 1.  Control flow (If / Loop: early exit, point pruning) and per-layer confidence heads are reported as problems:
    the kernels implement the fixed-depth graph."""
    inits, nodes = read_model(path)
    consts, producer = _graph_view(inits, nodes)
    hp = {k: None for k in Wt.LG_HPARAMS}
    problems = []
    idx = {int(k.split(".")[1]) for k in inits if k.startswith("transformers.") and k.split(".")[1].isdigit()}
    lins = _linears_in_order_of_use(inits, nodes)
    # constant folding gives every APPLICATION of a Linear its own anonymous copy of the weight (one per image side): count contents
    distinct = lambda shape: len({w.tobytes() for w, _, _ in lins if tuple(w.shape) == shape})
    n_qkv = distinct((768, 256))
    if idx:
        hp["layers"] = max(idx) + 1
        if n_qkv and n_qkv != hp["layers"]:
            problems.append(f"layers: parameter names say {hp['layers']}, the graph holds {n_qkv} Wqkv (768 x 256) Linears")
    elif n_qkv:
        hp["layers"] = n_qkv
    else:
        problems.append("layers: neither `transformers.{i}.` names nor 768 x 256 Linears found")
    heads = set()
    for n in nodes:
        if n["op"] != "Reshape" or len(n["inputs"]) < 2:
            continue
        shp = _const_of(n["inputs"][1], consts, producer)
        if shp is None:                       # shape assembled from dynamic dims: Concat of constants and Gather(Shape) pieces
            c = producer.get(n["inputs"][1])
            if c is not None and c["op"] == "Concat":
                parts = [_const_of(i, consts, producer) for i in c["inputs"]]
                shp = np.concatenate([np.asarray(p).reshape(-1) if p is not None else np.array([-7]) for p in parts])
        if shp is None:
            continue
        shp = [int(v) for v in np.asarray(shp).reshape(-1)]
        if len(shp) >= 4 and shp[-1] == 3 and shp[-3] > 0 and shp[-2] in (64, -1):
            heads.add(shp[-3])                 # Wqkv(x).unflatten(-1, (heads, -1, 3))
        elif len(shp) >= 4 and shp[-1] == 64 and shp[-2] > 0:
            heads.add(shp[-2])                 # to_qk / to_v: unflatten(-1, (heads, -1))
    if len(heads) == 1:
        hp["heads"] = heads.pop()
    else:
        problems.append(f"heads: head-split Reshape constants give {sorted(heads)}")
    thr = set()
    for n in nodes:
        if n["op"] == "Greater" and len(n["inputs"]) == 2:
            v = _scalar(_const_of(n["inputs"][1], consts, producer))
            if isinstance(v, float) and 0.0 < v < 1.0:
                thr.add(float(np.float32(v)))
    if len(thr) == 1:
        hp["filter_threshold"] = thr.pop()
    else:
        problems.append(f"filter_threshold: expected exactly one Greater(x, c) with 0 < c < 1, found constants {sorted(thr)}")
    ctl = sorted({n["op"] for n in nodes if n["op"] in ("If", "Loop", "Scan")})
    if ctl:
        problems.append(f"control flow {ctl} in the graph (early exit / point pruning): only the fixed-depth export is implemented")
    n_conf = distinct((1, 256))
    if n_conf > 1:
        problems.append(f"{n_conf} Linear(256 -> 1) heads: per-layer token-confidence heads of the early-exit graph are present "
                        "(the fixed-depth export keeps the last matchability head only)")
    return hp, problems


def resolve_hparams(kind, read, problems, assume=None, path="<model>"):
    """Merge what the graph says with what the caller states explicitly.  A value that could not be read AND is not assumed, a
    stated value that contradicts a read one, or a structural problem (anything in `problems` not about an assumable key)
    raises ValueError: a converted weight file never carries guessed hyper-parameters."""
    assume = dict(assume or {})
    keys = list(Wt.SP_HPARAMS if kind == 1 else Wt.LG_HPARAMS)
    unknown = set(assume) - set(keys)
    if unknown:
        raise ValueError(f"assume: unknown hyper-parameter(s) {sorted(unknown)}; known: {keys}")
    out, errors = {}, []
    for k in keys:
        if read.get(k) is not None and k in assume and float(assume[k]) != float(read[k]):
            errors.append(f"{k}: the graph says {read[k]}, assume says {assume[k]}")
        out[k] = read[k] if read.get(k) is not None else assume.get(k)
    for pmsg in problems:
        key = pmsg.split(":", 1)[0].strip()
        if key in keys and out.get(key) is not None and read.get(key) is None:
            continue                          # unreadable but stated by the caller
        errors.append(pmsg)
    if kind == 2 and not errors and (out["layers"] != Wt.LG_LAYERS or out["heads"] != 4):
        errors.append(f"LightGlue with {out['layers']} layers of {out['heads']} heads: the kernels are built for {Wt.LG_LAYERS} layers of 4 heads x 64")
    if kind == 1 and out.get("topk_always") is None and out.get("max_keypoints") is not None:
        out["topk_always"] = 0
        errors = [e for e in errors if not e.startswith("topk_always")]
    if kind == 1 and not errors and not (1 <= out["nms_radius"] <= 8 and 0 <= out["remove_borders"] <= 64 and 1 <= out["max_keypoints"] <= 4096):
        errors.append(f"SuperPoint hyper-parameters outside the library's range (radius 1..8, border 0..64, keypoints 1..4096): {out}")
    if errors:
        raise ValueError(f"{path}: graph hyper-parameters refused -- " + "; ".join(errors))
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


def convert(path, kind, assume=None):
    """-> (blob, hparams): weights in the canonical layout AND the graph's hyper-parameters (resolve_hparams: refused when they cannot
    be read and are not stated).  What `main` writes into an RFEW v2 container."""
    if kind == 1:
        hp = resolve_hparams(1, *read_superpoint_hparams(path), assume=assume, path=path)
        return convert_superpoint(path), hp
    hp = resolve_hparams(2, *read_lightglue_hparams(path), assume=assume, path=path)
    return convert_lightglue(path), hp


def _parse_assume(items):
    out = {}
    for it in items or []:
        k, _, v = it.partition("=")
        out[k.strip()] = float(v) if ("." in v or "e" in v.lower()) else int(v)
    return out


def main(argv=None):
    import argparse
    import os
    ap = argparse.ArgumentParser(description="convert superpoint.onnx / lightglue_sim.onnx to RFEW v2 containers: initializers -> canonical "
                                             "weight blob, graph constants -> hyper-parameter header")
    ap.add_argument("--superpoint"); ap.add_argument("--lightglue"); ap.add_argument("--out-dir", default=".")
    ap.add_argument("--assume-sp", action="append", metavar="KEY=VALUE",
                    help="state a SuperPoint hyper-parameter the graph does not reveal (max_keypoints, detection_threshold, nms_radius, remove_borders)")
    ap.add_argument("--assume-lg", action="append", metavar="KEY=VALUE", help="the same for LightGlue (layers, heads, filter_threshold)")
    a = ap.parse_args(argv)
    for path, kind, name, assume in ((a.superpoint, 1, "superpoint.rfew", a.assume_sp), (a.lightglue, 2, "lightglue_sim.rfew", a.assume_lg)):
        if not path:
            continue
        read, problems = (read_superpoint_hparams if kind == 1 else read_lightglue_hparams)(path)
        print(f"{path}: hyper-parameters read from the graph: {read}" + (f"; unresolved: {problems}" if problems else ""))
        blob, hp = convert(path, kind, _parse_assume(assume))
        Wt.save(os.path.join(a.out_dir, name), blob, kind, hp)
        print(f"  -> {os.path.join(a.out_dir, name)} (RFEW v2, {hp})")


if __name__ == "__main__":
    main()
