"""Torch modules shaped like the PUBLISHED SuperPoint / LightGlue, and their export through PyTorch's own C++ ONNX serialiser.

Build-container / test infrastructure only (nothing here is shipped or read by the product).  The reference's two model files
(onnxmodel/superpoint.onnx, src/Extractors/SPextractor.cc:93; onnxmodel/lightglue_sim.onnx, src/Matchers/lightglue_onnx.cpp:38) are
missing (.MISSING_LARGE_BLOBS:4-5); the closest thing this image allows is a graph FILE written by a real exporter from the published
module code with this repo's seeded weights and the tensor names the reference binds (`image` -> `keypoints` / `scores` /
`descriptors`, src/Extractors/superpoint_onnx.cc:100,133-134; `kpts0/kpts1/desc0/desc1` -> `matches0` / `mscores0`,
src/Matchers/lightglue_onnx.cpp:168-172,210-211).  Used by tests/test_onnx_exporter.py, tests/test_onnx_hparams.py,
tools/ort_parity.py's self-test and tools/gen_onnx_golden.py (graph-execution fixtures through tools/mini_onnx.py).

SuperPointPublished: layer list / names of the published SuperPoint (reference include/SuperPoint.h:24-41), dense maps only.
SuperPointWithTail: + the published tail (simple_nms = 5 x max_pool2d(2r+1, 1, r), border = -1, threshold, top-k,
grid_sample(bilinear, align_corners) + normalisation).  LightGluePublished: cvg naming `transformers.{i}.self_attn.Wqkv`,
interleaved qkv rows, `log_assignment.{i}`, the match filter in the graph."""
import os
import sys

import numpy as np
import torch

nn = torch.nn
F = torch.nn.functional

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from rover_slam_amd import weights as Wt  # noqa: E402

SETTINGS = [dict(max_keypoints=1024, detection_threshold=0.0005, nms_radius=4, remove_borders=4),
            dict(max_keypoints=2048, detection_threshold=0.005, nms_radius=3, remove_borders=2)]


class ExporterUnavailable(RuntimeError):
    pass


def serialise(model, args, input_names, output_names, fold, dynamic_axes=None):
    """ModelProto bytes from torch's own C++ serialiser, or skip when this torch build lacks the internals."""
    try:
        import importlib
        from torch.onnx._internal.torchscript_exporter import utils as U
        from torch.onnx._internal.torchscript_exporter._globals import GLOBALS
        from torch.onnx import OperatorExportTypes
        for v in range(9, 18):      # the symbolic functions register on import: without opset >= 11 in-place slice assignment has no export
            importlib.import_module(f"torch.onnx._internal.torchscript_exporter.symbolic_opset{v}")
        GLOBALS.export_onnx_opset_version = 17
    except Exception as e:                                    # pragma: no cover
        raise ExporterUnavailable(f"torch TorchScript ONNX exporter internals not available: {e}")
    with torch.no_grad():
        graph, params, _ = U._model_to_graph(model, args, do_constant_folding=fold, input_names=input_names,
                                             output_names=output_names, dynamic_axes=dynamic_axes or {})
        out = graph._export_onnx(params, 17, dynamic_axes or {}, False, OperatorExportTypes.ONNX, True, True, {}, True, "", {})
    proto = out[0]
    assert isinstance(proto, (bytes, bytearray)) and len(proto) > 1000
    return bytes(proto)



# ---------------------------------------------------------------------------------------------- published SuperPoint
class SuperPointPublished(nn.Module):
    """Layer list / names of the published SuperPoint (reference include/SuperPoint.h:24-41); forward up to the dense maps."""

    def __init__(self):
        super().__init__()
        for name, cin, cout, k in Wt.SP_LAYERS:
            setattr(self, name, nn.Conv2d(cin, cout, k, padding=k // 2))

    def forward(self, image):
        r, pool = torch.relu, lambda t: torch.nn.functional.max_pool2d(t, 2, 2)
        x = r(self.conv1a(image)); x = pool(r(self.conv1b(x)))
        x = r(self.conv2a(x)); x = pool(r(self.conv2b(x)))
        x = r(self.conv3a(x)); x = pool(r(self.conv3b(x)))
        x = r(self.conv4a(x)); x = r(self.conv4b(x))
        s = torch.softmax(self.convPb(r(self.convPa(x))), 1)[:, :-1]
        b, _, h, w = s.shape
        scores = s.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
        d = self.convDb(r(self.convDa(x)))
        desc = torch.nn.functional.normalize(d, p=2, dim=1)
        kp = torch.nonzero(scores[0] > 0.5)                   # stands in for the NMS / top-k tail (int64 keypoints output)
        return kp, scores, desc


def load_sp(m, blob):
    man, _ = Wt.sp_manifest()
    with torch.no_grad():
        for name, off, shape in man:
            layer, leaf = name.split(".")
            getattr(getattr(m, layer), leaf).copy_(torch.from_numpy(blob[off:off + int(np.prod(shape))].reshape(shape).copy()))
    return m.eval()


def simple_nms(scores, r):
    """published SuperPoint / LightGlue simple_nms"""
    mp = lambda x: F.max_pool2d(x, kernel_size=r * 2 + 1, stride=1, padding=r)
    zeros = torch.zeros_like(scores)
    max_mask = scores == mp(scores)
    for _ in range(2):
        supp_mask = mp(max_mask.float()) > 0
        supp_scores = torch.where(supp_mask, zeros, scores)
        new_max_mask = supp_scores == mp(supp_scores)
        max_mask = max_mask | (new_max_mask & (~supp_mask))
    return torch.where(max_mask, scores, zeros)


def sp_tail(scores, dmap, hp, topk="min"):
    """published tail on a [1,H,W] score map and a [1,256,Hc,Wc] normalised descriptor map -> keypoints (x, y) i64 [K,2], scores [K],
    descriptors [K,256].  topk = "min": torch.topk(scores, min(k, n)) as trace-friendly exports write it; "const": a constant k
    (the published top_k_keypoints when more than k candidates exist at trace time)."""
    scores = simple_nms(scores, hp["nms_radius"])
    pad = hp["remove_borders"]
    if pad > 0:
        scores[:, :pad] = -1
        scores[:, :, :pad] = -1
        scores[:, -pad:] = -1
        scores[:, :, -pad:] = -1
    best = torch.where(scores > hp["detection_threshold"])
    sc = scores[best]
    kp = torch.stack(best[1:3], dim=-1)
    if topk == "min":
        k = torch.minimum(torch.tensor(hp["max_keypoints"]), torch.tensor(sc.shape[0]))
    else:
        k = hp["max_keypoints"]
    sc, idx = torch.topk(sc, k, dim=0)
    kp = torch.flip(kp[idx], [1]).float()                                      # (y, x) -> (x, y)
    c = dmap.shape[1]
    # published: keypoints / torch.tensor([w * s - s / 2 - 0.5, h * s - s / 2 - 0.5]); written on the Shape tensor here because a
    # traced torch.tensor([...]) of Python ints freezes the TRACE-TIME image size into the graph (found by executing the exported file
    # at another size through tools/mini_onnx.py: descriptors off by 0.25 while the torch module, re-evaluating .shape, agreed)
    shp = torch._shape_as_tensor(dmap)
    g = (kp - 8 / 2 + 0.5) / (torch.stack([shp[3], shp[2]]).float() * 8 - (8 / 2 + 0.5))
    d = F.grid_sample(dmap, (g * 2 - 1).view(1, 1, -1, 2), mode="bilinear", align_corners=True)
    d = F.normalize(d.reshape(1, c, -1), p=2, dim=1)
    return kp.long(), sc, d[0].transpose(0, 1)


class SuperPointWithTail(SuperPointPublished):
    def __init__(self, hp, topk="min"):
        super().__init__()
        self.hp, self.topk = hp, topk

    def forward(self, image):
        r, pool = torch.relu, lambda t: F.max_pool2d(t, 2, 2)
        x = r(self.conv1a(image)); x = pool(r(self.conv1b(x)))
        x = r(self.conv2a(x)); x = pool(r(self.conv2b(x)))
        x = r(self.conv3a(x)); x = pool(r(self.conv3b(x)))
        x = r(self.conv4a(x)); x = r(self.conv4b(x))
        s = torch.softmax(self.convPb(r(self.convPa(x))), 1)[:, :-1]
        b, _, h, w = s.shape
        scores = s.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
        dmap = F.normalize(self.convDb(r(self.convDa(x))), p=2, dim=1)
        kp, sc, d = sp_tail(scores, dmap, self.hp, self.topk)
        return kp[None], sc[None], d[None]


def export_sp(tmp_path, hp, topk="min", seed=5, size=(64, 80), dustbin_bias=0.0, desc_center=None):
    """superpoint.onnx as the reference names it, dynamic image size; returns (path, canonical weight blob)"""
    blob = Wt.make_superpoint(seed=seed, dustbin_bias=dustbin_bias, desc_center=desc_center)
    m = load_sp(SuperPointWithTail(hp, topk), blob)
    proto = serialise(m, (torch.rand(1, 1, *size),), ["image"], ["keypoints", "scores", "descriptors"], True, {"image": {2: "h", 3: "w"}})
    path = os.path.join(str(tmp_path), "superpoint.onnx")
    with open(path, "wb") as f:
        f.write(proto)
    return path, blob



# ---------------------------------------------------------------------------------------------- published LightGlue
def _rotate_half(x):          # published: x.unflatten(-1, (-1, 2)); spelled with reshape like the ONNX-exportable forks (TorchScript has no unflatten)
    x = x.reshape(x.shape[:-1] + (-1, 2))
    x1, x2 = x[..., 0], x[..., 1]
    return torch.stack((-x2, x1), dim=-1).flatten(start_dim=-2)


def _apply_rotary(freqs, t):
    return t * freqs[0] + _rotate_half(t) * freqs[1]


class _PosEnc(nn.Module):
    def __init__(self):
        super().__init__()
        self.Wr = nn.Linear(2, 32, bias=False)

    def forward(self, x):
        p = self.Wr(x)
        emb = torch.stack([torch.cos(p), torch.sin(p)], 0).unsqueeze(2)       # published: .unsqueeze(-3)
        return torch.stack([emb, emb], -1).flatten(4)   # published: emb.repeat_interleave(2, dim=-1) (the TorchScript exporter mis-infers its rank)


def _ffn():
    return nn.Sequential(nn.Linear(512, 512), nn.LayerNorm(512, elementwise_affine=True), nn.GELU(), nn.Linear(512, 256))


def _attend(q, k, v):
    s = torch.matmul(q, k.transpose(2, 3)) * 0.125          # [b, heads, n, 64]; 64 ** -0.5
    return torch.matmul(torch.softmax(s, -1), v)


class _SelfBlock(nn.Module):
    def __init__(self):
        super().__init__()
        self.Wqkv = nn.Linear(256, 768)
        self.out_proj = nn.Linear(256, 256)
        self.ffn = _ffn()

    def forward(self, x, enc):
        y = self.Wqkv(x)
        qkv = y.reshape(y.shape[0], y.shape[1], 4, 64, 3).transpose(1, 2)     # published: .unflatten(-1, (heads, -1, 3)): interleaved (head, dim, q|k|v)
        q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
        ctx = _attend(_apply_rotary(enc, q), _apply_rotary(enc, k), v)
        msg = self.out_proj(ctx.transpose(1, 2).flatten(start_dim=-2))
        return x + self.ffn(torch.cat([x, msg], -1))


class _CrossBlock(nn.Module):
    def __init__(self):
        super().__init__()
        self.to_qk = nn.Linear(256, 256)
        self.to_v = nn.Linear(256, 256)
        self.to_out = nn.Linear(256, 256)
        self.ffn = _ffn()

    def forward(self, x0, x1):
        hd = lambda t: t.reshape(t.shape[0], t.shape[1], 4, 64).transpose(1, 2)
        qk0, qk1, v0, v1 = hd(self.to_qk(x0)), hd(self.to_qk(x1)), hd(self.to_v(x0)), hd(self.to_v(x1))
        m0 = self.to_out(_attend(qk0, qk1, v1).transpose(1, 2).flatten(start_dim=-2))
        m1 = self.to_out(_attend(qk1, qk0, v0).transpose(1, 2).flatten(start_dim=-2))
        return x0 + self.ffn(torch.cat([x0, m0], -1)), x1 + self.ffn(torch.cat([x1, m1], -1))


class _Layer(nn.Module):
    def __init__(self):
        super().__init__()
        self.self_attn = _SelfBlock()
        self.cross_attn = _CrossBlock()

    def forward(self, d0, d1, e0, e1):
        return self.cross_attn(self.self_attn(d0, e0), self.self_attn(d1, e1))


class _Assign(nn.Module):
    def __init__(self):
        super().__init__()
        self.matchability = nn.Linear(256, 1)
        self.final_proj = nn.Linear(256, 256)

    def forward(self, d0, d1):
        md0, md1 = self.final_proj(d0) / 256 ** 0.25, self.final_proj(d1) / 256 ** 0.25
        sim = torch.matmul(md0, md1.transpose(1, 2))
        ls = torch.nn.functional.logsigmoid
        z0, z1 = self.matchability(d0), self.matchability(d1)
        return torch.log_softmax(sim, 2) + torch.log_softmax(sim, 1) + ls(z0) + ls(z1).transpose(1, 2)


class LightGluePublished(nn.Module):
    """Module tree / parameter names of the published LightGlue (input_proj = identity at 256-d SuperPoint descriptors); the fused
    export has no early exit, so only the last log_assignment head is live (the others never reach the file)."""

    def __init__(self, n_layers=Wt.LG_LAYERS, filter_threshold=0.1):
        super().__init__()
        self.filter_threshold = filter_threshold
        self.posenc = _PosEnc()
        self.transformers = nn.ModuleList([_Layer() for _ in range(n_layers)])
        self.log_assignment = nn.ModuleList([_Assign() for _ in range(n_layers)])

    def forward(self, kpts0, kpts1, desc0, desc1):
        e0, e1 = self.posenc(kpts0), self.posenc(kpts1)
        d0, d1 = desc0, desc1
        for layer in self.transformers:
            d0, d1 = layer(d0, d1, e0, e1)
        scores = self.log_assignment[-1](d0, d1)
        m0 = scores.max(2)
        m1 = scores.max(1)
        idx = torch.arange(scores.shape[1])[None]
        mutual = m1.indices.gather(1, m0.indices) == idx
        ms = torch.where(mutual, m0.values.exp(), torch.zeros_like(m0.values))
        valid = ms[0] > self.filter_threshold
        i = torch.nonzero(valid)[:, 0]
        return torch.stack([i, m0.indices[0][i]], -1), ms[0][i], d0, d1, scores


def interleave_qkv(w, b):
    """canonical rows t*256 + h*64 + d  ->  published rows h*192 + d*3 + t (inverse of onnx_weights._deinterleave_qkv)"""
    return (np.ascontiguousarray(w.reshape(3, 4, 64, 256).transpose(1, 2, 0, 3).reshape(768, 256)),
            np.ascontiguousarray(b.reshape(3, 4, 64).transpose(1, 2, 0).reshape(768)))


def load_lg(m, blob):
    man, _ = Wt.lg_manifest()
    t = {name: blob[off:off + int(np.prod(shape))].reshape(shape).copy() for name, off, shape in man}
    cp = lambda p, a: p.copy_(torch.from_numpy(np.ascontiguousarray(a)))
    with torch.no_grad():
        cp(m.posenc.Wr.weight, t["posenc.Wr"])
        for l, L in enumerate(m.transformers):
            p, s, c = f"layers.{l}.", L.self_attn, L.cross_attn
            w, b = interleave_qkv(t[p + "self.Wqkv"], t[p + "self.bqkv"])
            cp(s.Wqkv.weight, w); cp(s.Wqkv.bias, b)
            for mod, wn, bn in ((s.out_proj, "self.Wo", "self.bo"), (s.ffn[0], "self.W1", "self.b1"), (s.ffn[1], "self.ln_g", "self.ln_b"),
                                (s.ffn[3], "self.W2", "self.b2"), (c.to_qk, "cross.Wqk", "cross.bqk"), (c.to_v, "cross.Wv", "cross.bv"),
                                (c.to_out, "cross.Wo", "cross.bo"), (c.ffn[0], "cross.W1", "cross.b1"), (c.ffn[1], "cross.ln_g", "cross.ln_b"),
                                (c.ffn[3], "cross.W2", "cross.b2")):
                cp(mod.weight, t[p + wn]); cp(mod.bias, t[p + bn])
        for a in m.log_assignment[:-1]:                     # dead heads: anything but the live weights
            for q in a.parameters():
                q.fill_(0.5)
        a = m.log_assignment[-1]
        cp(a.final_proj.weight, t["final_proj.W"]); cp(a.final_proj.bias, t["final_proj.b"])
        cp(a.matchability.weight, t["matchability.w"].reshape(1, 256)); cp(a.matchability.bias, t["matchability.b"])
    return m.eval()


def lg_inputs(n0, n1, seed):
    rng = np.random.default_rng(seed)
    d0 = rng.standard_normal((n0, 256)).astype(np.float32); d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    d1 = rng.standard_normal((n1, 256)).astype(np.float32); d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    m = min(n0, n1)
    d1[:m] = d0[:m] + 0.02 * rng.standard_normal((m, 256)).astype(np.float32)
    d1[:m] /= np.linalg.norm(d1[:m], axis=1, keepdims=True)
    k0 = rng.uniform(-0.9, 0.9, (n0, 2)).astype(np.float32)
    k1 = rng.uniform(-0.9, 0.9, (n1, 2)).astype(np.float32); k1[:m] = k0[:m] + 0.01
    return k0, k1, d0, d1


def export_lg(tmp_path, filter_threshold=0.1, seed=3, fold=True, n_layers=Wt.LG_LAYERS, trace_sizes=(12, 9), calibrated=False):
    """lightglue_sim.onnx as the reference names it, dynamic keypoint counts; returns (path, canonical weight blob)"""
    blob = Wt.make_lightglue(seed=seed, calibrated=calibrated)
    m = load_lg(LightGluePublished(n_layers, filter_threshold), blob) if n_layers == Wt.LG_LAYERS else LightGluePublished(n_layers, filter_threshold).eval()
    k0, k1, d0, d1 = (torch.from_numpy(a)[None] for a in lg_inputs(*trace_sizes, 0))
    names = ["kpts0", "kpts1", "desc0", "desc1"]
    proto = serialise(m, (k0, k1, d0, d1), names, ["matches0", "mscores0"], fold, {n: {1: "n" + n[-1]} for n in names})
    path = os.path.join(str(tmp_path), "lightglue_sim.onnx")
    with open(path, "wb") as f:
        f.write(proto)
    return path, blob
