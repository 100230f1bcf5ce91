#!/usr/bin/env python3
"""A minimal ONNX graph interpreter with an onnxruntime-shaped front (`InferenceSession(path).run(names, feeds)`).

BUILD-CONTAINER / TEST INFRASTRUCTURE ONLY -- never shipped, never imported by the product, not a performance path.

Why it exists: the reference's arithmetic is `Ort::Session::Run` on two graph files
(src/Extractors/superpoint_onnx.cc:133-136, src/Matchers/lightglue_onnx.cpp:210-214); neither onnxruntime nor the `onnx` package is
in this image.  The oracle was pinned against the torch MODULES the test exports are traced from, never against an EXECUTION OF THE
EXPORTED GRAPH FILE the converter ingests.  This interpreter closes that gap as far as the image allows: it walks the node list
`rover_slam_amd.onnx_weights.read_model` parses from the file (the same wire-format reader the converter uses) and evaluates every
node with numpy / torch-CPU primitives following the ONNX operator specification (opset 11-18 forms of the ~55 operators the
SuperPoint-with-tail and fused LightGlue exports use), so that a TopK tie rule, the (y, x) -> (x, y) flip, the ScatterND border, an
int64 cast or a Slice bound that differs between module and graph shows up as a deviation of `tools/ort_parity.py --backend mini`.

Heavy fp32 operators (Conv, MaxPool, MatMul, Gemm, Softmax, LogSoftmax, LayerNormalization, Erf, GridSample) run on torch-CPU
kernels; everything else is numpy.  TopK follows the specification's tie rule (equal values: lower index first), as onnxruntime does.
Unknown operators raise NotImplementedError naming the node -- nothing is skipped silently.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from rover_slam_amd import onnx_weights as OW  # noqa: E402

_ONNX_DT = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64,
            12: np.uint32, 13: np.uint64}


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a))


def _axes(node, ins, pos=1, key="axes"):
    """axes: an input from opset 13 (Squeeze / Unsqueeze / ReduceSum) or 18 (the other reductions), an attribute before"""
    if len(ins) > pos and ins[pos] is not None:
        return [int(v) for v in np.asarray(ins[pos]).reshape(-1)]
    a = node["attrs"].get(key)
    if a is None:
        return None
    return [int(a)] if isinstance(a, int) else [int(v) for v in a]


def _int_div(a, b):
    if np.issubdtype(np.asarray(a).dtype, np.integer):       # ONNX integer Div truncates toward zero (C semantics), numpy's // floors
        q = np.abs(a) // np.abs(b)
        return (q * (np.sign(a) * np.sign(b))).astype(np.result_type(a, b))
    return a / b


def _reduce(fn):
    def op(node, ins):
        x = ins[0]
        ax = _axes(node, ins)
        keep = bool(node["attrs"].get("keepdims", 1))
        if ax is None and node["attrs"].get("noop_with_empty_axes", 0):
            return [x]
        return [np.asarray(fn(x, tuple(ax) if ax is not None else None, keep)).astype(x.dtype)]
    return op


def _conv(node, ins):
    import torch
    a = node["attrs"]
    x, w = _t(ins[0]), _t(ins[1])
    b = _t(ins[2]) if len(ins) > 2 and ins[2] is not None else None
    nd = w.dim() - 2
    if a.get("auto_pad", "NOTSET") not in ("NOTSET", ""):
        raise NotImplementedError(f"Conv auto_pad = {a['auto_pad']}")
    pads = list(a.get("pads") or [0] * (2 * nd))
    if pads[:nd] != pads[nd:]:
        x = torch.nn.functional.pad(x, [p for d in reversed(range(nd)) for p in (pads[d], pads[nd + d])])
        pads = [0] * (2 * nd)
    fn = {1: torch.nn.functional.conv1d, 2: torch.nn.functional.conv2d}[nd]
    y = fn(x, w, b, stride=list(a.get("strides") or [1] * nd), padding=pads[:nd], dilation=list(a.get("dilations") or [1] * nd),
           groups=int(a.get("group", 1)))
    return [y.numpy()]


def _maxpool(node, ins):
    import torch
    a = node["attrs"]
    ks = list(a["kernel_shape"])
    nd = len(ks)
    pads = list(a.get("pads") or [0] * (2 * nd))
    if pads[:nd] != pads[nd:] or nd != 2 or a.get("auto_pad", "NOTSET") not in ("NOTSET", "") or a.get("storage_order", 0):
        raise NotImplementedError(f"MaxPool form {a}")
    y = torch.nn.functional.max_pool2d(_t(ins[0]), ks, stride=list(a.get("strides") or [1] * nd), padding=pads[:nd],
                                       dilation=list(a.get("dilations") or [1] * nd), ceil_mode=bool(a.get("ceil_mode", 0)))
    return [y.numpy()]


def _slice(node, ins):
    x = ins[0]
    if len(ins) > 1:
        starts, ends = np.asarray(ins[1]).reshape(-1), np.asarray(ins[2]).reshape(-1)
        axes = np.asarray(ins[3]).reshape(-1) if len(ins) > 3 and ins[3] is not None else np.arange(len(starts))
        steps = np.asarray(ins[4]).reshape(-1) if len(ins) > 4 and ins[4] is not None else np.ones(len(starts), np.int64)
    else:                                                       # opset < 10: attributes
        starts, ends = node["attrs"]["starts"], node["attrs"]["ends"]
        axes = node["attrs"].get("axes", list(range(len(starts))))
        steps = [1] * len(starts)
    sl = [slice(None)] * x.ndim
    for s, e, ax, st in zip(starts, ends, axes, steps):
        s, e, ax, st = int(s), int(e), int(ax), int(st)
        if st == 0:
            raise ValueError("Slice: step 0")
        n = x.shape[ax]
        # the specification's clamping: negative values count from the end, then clamp to [0, n] (step > 0) or [-1, n - 1] (step < 0)
        s = s + n if s < 0 else s
        e = e + n if e < 0 else e
        if st > 0:
            s, e = min(max(s, 0), n), min(max(e, 0), n)
            sl[ax] = slice(s, e, st)
        else:
            s, e = min(max(s, 0), n - 1), min(max(e, -1), n - 1)
            sl[ax] = slice(s, e if e >= 0 else None, st)
    return [x[tuple(sl)]]


def _reshape(node, ins):
    x, shp = ins[0], [int(v) for v in np.asarray(ins[1]).reshape(-1)]
    if not node["attrs"].get("allowzero", 0):
        shp = [x.shape[i] if v == 0 else v for i, v in enumerate(shp)]
    return [x.reshape(shp)]


def _scatter_nd(node, ins):
    if node["attrs"].get("reduction", "none") not in ("none", ""):
        raise NotImplementedError("ScatterND with a reduction")
    data, idx, upd = ins[0].copy(), np.asarray(ins[1]), np.asarray(ins[2])
    k = idx.shape[-1]
    flat = idx.reshape(-1, k)
    flat = np.where(flat < 0, flat + np.array(data.shape[:k]), flat)
    data[tuple(flat.T)] = upd.reshape((flat.shape[0],) + data.shape[k:])
    return [data]


def _topk(node, ins):
    a = node["attrs"]
    x = ins[0]
    k = int(np.asarray(ins[1]).reshape(-1)[0]) if len(ins) > 1 else int(a["k"])
    ax = int(a.get("axis", -1))
    # specification: "given two equivalent values, the element with the lower index appears first" -> a STABLE sort on the key
    key = -x if a.get("largest", 1) else x
    if x.dtype == np.bool_ or (np.issubdtype(x.dtype, np.integer) and a.get("largest", 1)):
        key = -(x.astype(np.int64))
    order = np.argsort(key, axis=ax, kind="stable")
    idx = np.take(order, np.arange(k), axis=ax)
    return [np.take_along_axis(x, idx, axis=ax), idx.astype(np.int64)]


def _split(node, ins):
    x, a = ins[0], node["attrs"]
    ax = int(a.get("axis", 0))
    if len(ins) > 1 and ins[1] is not None:
        sizes = [int(v) for v in np.asarray(ins[1]).reshape(-1)]
    elif "split" in a:
        sizes = list(a["split"])
    else:
        n = len(node["outputs"])
        sizes = [-(-x.shape[ax] // n)] * n
        sizes[-1] = x.shape[ax] - sum(sizes[:-1])
    return [np.ascontiguousarray(p) for p in np.split(x, np.cumsum(sizes)[:-1], axis=ax)]


def _grid_sample(node, ins):
    import torch
    a = node["attrs"]
    mode = {"linear": "bilinear", "cubic": "bicubic"}.get(a.get("mode", "bilinear"), a.get("mode", "bilinear"))
    y = torch.nn.functional.grid_sample(_t(ins[0]), _t(ins[1]), mode=mode, padding_mode=a.get("padding_mode", "zeros"),
                                        align_corners=bool(a.get("align_corners", 0)))
    return [y.numpy()]


def _layernorm(node, ins):
    import torch
    x = _t(ins[0])
    ax = int(node["attrs"].get("axis", -1)) % x.dim()
    y = torch.nn.functional.layer_norm(x, tuple(x.shape[ax:]), _t(ins[1]) if len(ins) > 1 and ins[1] is not None else None,
                                       _t(ins[2]) if len(ins) > 2 and ins[2] is not None else None, float(node["attrs"].get("epsilon", 1e-5)))
    return [y.numpy()]


def _gemm(node, ins):
    import torch
    a = node["attrs"]
    A, B = _t(ins[0]), _t(ins[1])
    A = A.t() if a.get("transA", 0) else A
    B = B.t() if a.get("transB", 0) else B
    y = float(a.get("alpha", 1.0)) * torch.matmul(A, B) if float(a.get("alpha", 1.0)) != 1.0 else torch.matmul(A, B)
    if len(ins) > 2 and ins[2] is not None:
        c = _t(ins[2])
        y = y + (float(a.get("beta", 1.0)) * c if float(a.get("beta", 1.0)) != 1.0 else c)
    return [y.numpy()]


def _softmax(log):
    def op(node, ins):
        import torch
        x = _t(ins[0])
        ax = int(node["attrs"].get("axis", -1))
        return [(torch.log_softmax if log else torch.softmax)(x, ax).numpy()]
    return op


def _torch1(name):
    def op(node, ins):
        import torch
        return [getattr(torch, name)(_t(ins[0])).numpy()]
    return op


def _constant(node, ins):
    a = node["attrs"]
    for key in ("value", "value_float", "value_int", "value_floats", "value_ints"):
        if key in a and not isinstance(a[key], str):
            v = a[key]
            if key == "value_float":
                return [np.asarray(v, np.float32)]
            if key == "value_int":
                return [np.asarray(v, np.int64)]
            if key == "value_floats":
                return [np.asarray(v, np.float32)]
            if key == "value_ints":
                return [np.asarray(v, np.int64)]
            return [np.asarray(v)]
    raise NotImplementedError(f"Constant node {node['name']} with attributes {sorted(a)}")


def _constant_of_shape(node, ins):
    v = node["attrs"].get("value")
    v = np.zeros((), np.float32) if v is None else np.asarray(v).reshape(-1)[0]
    return [np.full([int(d) for d in np.asarray(ins[0]).reshape(-1)], v, dtype=np.asarray(v).dtype)]


def _shape(node, ins):
    s = np.array(ins[0].shape, np.int64)
    st, en = node["attrs"].get("start", 0), node["attrs"].get("end")
    return [s[st:en]]


def _squeeze(node, ins):
    ax = _axes(node, ins)
    return [np.squeeze(ins[0], tuple(ax)) if ax is not None else np.squeeze(ins[0])]


def _unsqueeze(node, ins):
    x = ins[0]
    ax = _axes(node, ins)
    rank = x.ndim + len(ax)
    for d in sorted(a % rank for a in ax):
        x = np.expand_dims(x, d)
    return [x]


def _cast(node, ins):
    to = int(node["attrs"]["to"])
    if to not in _ONNX_DT:
        raise NotImplementedError(f"Cast to ONNX type {to}")
    return [ins[0].astype(_ONNX_DT[to])]


def _clip(node, ins):
    lo = ins[1] if len(ins) > 1 and ins[1] is not None else node["attrs"].get("min")
    hi = ins[2] if len(ins) > 2 and ins[2] is not None else node["attrs"].get("max")
    x = ins[0]
    if lo is not None:
        x = np.maximum(x, np.asarray(lo, x.dtype))
    if hi is not None:
        x = np.minimum(x, np.asarray(hi, x.dtype))
    return [x]


def _argm(fn):
    def op(node, ins):
        a = node["attrs"]
        if a.get("select_last_index", 0):
            raise NotImplementedError("ArgMax/ArgMin select_last_index")
        ax = int(a.get("axis", 0))
        r = fn(ins[0], axis=ax).astype(np.int64)               # first occurrence of the extremum, as the specification's default
        return [np.expand_dims(r, ax) if a.get("keepdims", 1) else r]
    return op


def _variadic(fn):
    def op(node, ins):
        r = ins[0]
        for x in ins[1:]:
            r = fn(r, x)
        return [r]
    return op


_OPS = {
    "Constant": _constant, "ConstantOfShape": _constant_of_shape, "Shape": _shape, "Identity": lambda n, i: [i[0]],
    "Cast": _cast, "Reshape": _reshape, "Squeeze": _squeeze, "Unsqueeze": _unsqueeze, "Slice": _slice, "Split": _split,
    "Flatten": lambda n, i: [i[0].reshape(int(np.prod(i[0].shape[:int(n["attrs"].get("axis", 1))], dtype=np.int64)), -1)],
    "Transpose": lambda n, i: [np.transpose(i[0], n["attrs"].get("perm"))],
    "Concat": lambda n, i: [np.concatenate(list(i), axis=int(n["attrs"]["axis"]))],
    "Expand": lambda n, i: [np.broadcast_to(i[0], np.broadcast_shapes(i[0].shape, tuple(int(d) for d in np.asarray(i[1]).reshape(-1)))).copy()],
    "Gather": lambda n, i: [np.take(i[0], np.asarray(i[1]), axis=int(n["attrs"].get("axis", 0)))],
    "GatherElements": lambda n, i: [np.take_along_axis(i[0], np.where(i[1] < 0, i[1] + i[0].shape[int(n["attrs"].get("axis", 0))], i[1]),
                                                       axis=int(n["attrs"].get("axis", 0)))],
    "ScatterND": _scatter_nd, "NonZero": lambda n, i: [np.array(np.nonzero(i[0]), np.int64).reshape(i[0].ndim, -1)],
    "Range": lambda n, i: [np.arange(np.asarray(i[0]).item(), np.asarray(i[1]).item(), np.asarray(i[2]).item(), dtype=np.asarray(i[0]).dtype)],
    "TopK": _topk, "ArgMax": _argm(np.argmax), "ArgMin": _argm(np.argmin),
    "Add": lambda n, i: [i[0] + i[1]], "Sub": lambda n, i: [i[0] - i[1]], "Mul": lambda n, i: [i[0] * i[1]], "Div": lambda n, i: [_int_div(i[0], i[1])],
    "Neg": lambda n, i: [-i[0]], "Abs": lambda n, i: [np.abs(i[0])], "Sqrt": lambda n, i: [np.sqrt(i[0])], "Pow": lambda n, i: [np.power(i[0], i[1]).astype(i[0].dtype)],
    "Reciprocal": lambda n, i: [np.reciprocal(i[0])], "Floor": lambda n, i: [np.floor(i[0])], "Ceil": lambda n, i: [np.ceil(i[0])],
    "Min": _variadic(np.minimum), "Max": _variadic(np.maximum), "Sum": _variadic(np.add), "Clip": _clip,
    "Equal": lambda n, i: [np.equal(i[0], i[1])], "Greater": lambda n, i: [np.greater(i[0], i[1])], "Less": lambda n, i: [np.less(i[0], i[1])],
    "GreaterOrEqual": lambda n, i: [np.greater_equal(i[0], i[1])], "LessOrEqual": lambda n, i: [np.less_equal(i[0], i[1])],
    "And": lambda n, i: [np.logical_and(i[0], i[1])], "Or": lambda n, i: [np.logical_or(i[0], i[1])], "Not": lambda n, i: [np.logical_not(i[0])],
    "Where": lambda n, i: [np.where(i[0], i[1], i[2])],
    "Relu": lambda n, i: [np.maximum(i[0], np.zeros((), i[0].dtype))], "Sigmoid": _torch1("sigmoid"), "Tanh": _torch1("tanh"), "Erf": _torch1("erf"),
    "Exp": _torch1("exp"), "Log": _torch1("log"), "Sin": _torch1("sin"), "Cos": _torch1("cos"),
    "Softmax": _softmax(False), "LogSoftmax": _softmax(True), "LayerNormalization": _layernorm,
    "Conv": _conv, "MaxPool": _maxpool, "GridSample": _grid_sample, "Gemm": _gemm,
    "MatMul": lambda n, i: [__import__("torch").matmul(_t(i[0]), _t(i[1])).numpy()],
    "ReduceMax": _reduce(lambda x, ax, k: np.max(x, axis=ax, keepdims=k)), "ReduceMin": _reduce(lambda x, ax, k: np.min(x, axis=ax, keepdims=k)),
    "ReduceSum": _reduce(lambda x, ax, k: np.sum(x, axis=ax, keepdims=k)), "ReduceMean": _reduce(lambda x, ax, k: np.mean(x, axis=ax, keepdims=k)),
    "ReduceL2": _reduce(lambda x, ax, k: np.sqrt(np.sum(x * x, axis=ax, keepdims=k))),
    "ReduceProd": _reduce(lambda x, ax, k: np.prod(x, axis=ax, keepdims=k)),
}
# Softmax / LogSoftmax before opset 13 flatten to 2-D around `axis` (default 1); handled in run() by refusing such files explicitly


class _Arg:
    def __init__(self, name):
        self.name = name


class InferenceSession:
    """The slice of onnxruntime.InferenceSession the parity harness uses: `get_inputs()`, `get_outputs()`, `run(names, feeds)`."""

    def __init__(self, path, sess_options=None, providers=None):
        self.path = path
        self.inits, self.nodes = OW.read_model(path)
        self.io = OW.read_graph_io(path)
        ops = {n["op"] for n in self.nodes}
        missing = sorted(ops - set(_OPS))
        if missing:
            raise NotImplementedError(f"{path}: operators not implemented by tools/mini_onnx.py: {missing}")
        if self.io["opset"] is not None and not 11 <= self.io["opset"] <= 18:
            raise NotImplementedError(f"{path}: opset {self.io['opset']}; the operator forms implemented here are those of opsets 11-18")
        if self.io["opset"] is not None and self.io["opset"] < 13 and ops & {"Softmax", "LogSoftmax"}:
            raise NotImplementedError(f"{path}: Softmax before opset 13 (flatten-to-2D semantics) is not implemented")
        self.op_counts = {}
        for n in self.nodes:
            self.op_counts[n["op"]] = self.op_counts.get(n["op"], 0) + 1

    def get_inputs(self):
        return [_Arg(n) for n in self.io["inputs"]]

    def get_outputs(self):
        return [_Arg(n) for n in self.io["outputs"]]

    def run(self, output_names, feeds):
        unknown = set(feeds) - set(self.io["inputs"])
        absent = set(self.io["inputs"]) - set(feeds)
        if unknown or absent:       # onnxruntime raises INVALID_ARGUMENT for both
            raise ValueError(f"{self.path}: feeds {sorted(feeds)} do not match the graph inputs {self.io['inputs']}")
        env = dict(self.inits)
        env.update({k: np.asarray(v) for k, v in feeds.items()})
        last_use = {}
        for idx, n in enumerate(self.nodes):
            for i in n["inputs"]:
                last_use[i] = idx
        keep = set(output_names or self.io["outputs"]) | set(self.inits)
        for idx, n in enumerate(self.nodes):     # ONNX requires the node list to be topologically sorted
            try:
                ins = [env[i] if i else None for i in n["inputs"]]
            except KeyError as e:
                raise ValueError(f"{self.path}: node {idx} {n['op']} reads {e} before anything produced it") from None
            while ins and ins[-1] is None:
                ins.pop()
            outs = _OPS[n["op"]](n, ins)
            for name, val in zip(n["outputs"], outs):
                if name:
                    env[name] = np.asarray(val)
            for i in n["inputs"]:                # free intermediates after their last consumer (SuperPoint at 480 x 640 keeps ~160 MB maps)
                if i and last_use.get(i) == idx and i not in keep and i in env:
                    del env[i]
        names = list(output_names or self.io["outputs"])
        missing = [o for o in names if o not in env]
        if missing:
            raise ValueError(f"{self.path}: no such output(s) {missing}; the graph's are {self.io['outputs']}")
        return [env[o] for o in names]


def get_available_providers():
    return ["MiniOnnxNumpyTorchCPU"]
