"""Seeded synthetic weights + the RFEW container the C ABI loads (rfe_load_weights).

The reference ships no weights in-tree (onnxmodel/*.onnx are missing, .MISSING_LARGE_BLOBS:4-5),
so every test / bench in this repo runs on synthetic, seeded, architecture-shaped weights.
Canonical blob layouts (float32, concatenated in this order):

SuperPoint (1 300 865 floats): for each of conv1a conv1b conv2a conv2b conv3a conv3b conv4a conv4b
  convPa convPb convDa convDb: weight [Cout,Cin,k,k] (PyTorch OIHW) then bias [Cout].
  (layer names: reference include/SuperPoint.h:24-41)

LightGlue: Wr [32,2]; 9 x { self: Wqkv [768,256] (rows 0..255 = q, 256..511 = k, 512..767 = v,
  each head-major 4x64), bqkv, Wo [256,256], bo, W1 [512,512], b1, ln_g [512], ln_b [512],
  W2 [256,512], b2;  cross: Wqk, bqk, Wv, bv, Wo, bo, W1, b1, ln_g, ln_b, W2, b2 };
  final_proj W [256,256], b; matchability w [256], b [1].
"""
import struct
import numpy as np

SP_LAYERS = [  # name, cin, cout, k
    ("conv1a", 1, 64, 3), ("conv1b", 64, 64, 3), ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3),
    ("conv3a", 64, 128, 3), ("conv3b", 128, 128, 3), ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3),
    ("convPa", 128, 256, 3), ("convPb", 256, 65, 1), ("convDa", 128, 256, 3), ("convDb", 256, 256, 1),
]
LG_LAYERS = 9


def sp_manifest():
    out, off = [], 0
    for name, cin, cout, k in SP_LAYERS:
        out.append((name + ".weight", off, (cout, cin, k, k)))
        off += cout * cin * k * k
        out.append((name + ".bias", off, (cout,)))
        off += cout
    return out, off


def lg_manifest():
    out, off = [], 0

    def take(name, shape):
        nonlocal off
        out.append((name, off, shape))
        off += int(np.prod(shape))

    take("posenc.Wr", (32, 2))
    for l in range(LG_LAYERS):
        p = f"layers.{l}."
        take(p + "self.Wqkv", (768, 256)); take(p + "self.bqkv", (768,))
        take(p + "self.Wo", (256, 256)); take(p + "self.bo", (256,))
        take(p + "self.W1", (512, 512)); take(p + "self.b1", (512,))
        take(p + "self.ln_g", (512,)); take(p + "self.ln_b", (512,))
        take(p + "self.W2", (256, 512)); take(p + "self.b2", (256,))
        take(p + "cross.Wqk", (256, 256)); take(p + "cross.bqk", (256,))
        take(p + "cross.Wv", (256, 256)); take(p + "cross.bv", (256,))
        take(p + "cross.Wo", (256, 256)); take(p + "cross.bo", (256,))
        take(p + "cross.W1", (512, 512)); take(p + "cross.b1", (512,))
        take(p + "cross.ln_g", (512,)); take(p + "cross.ln_b", (512,))
        take(p + "cross.W2", (256, 512)); take(p + "cross.b2", (256,))
    take("final_proj.W", (256, 256)); take("final_proj.b", (256,))
    take("matchability.w", (256,)); take("matchability.b", (1,))
    return out, off


SP_COUNT = sp_manifest()[1]
LG_COUNT = lg_manifest()[1]


def make_superpoint(seed=7, dustbin_bias=0.0, logit_gain=4.0, desc_center=None):
    """He-normal convs, biases U[-0.05,0.05].  `logit_gain` widens the 65-way logits so the score
    map is not near-uniform; `dustbin_bias` > 0 pushes mass into channel 64 (fewer keypoints).

    `desc_center`: None = the plain law (every round-1..4 fixture).  "auto" or a [256] vector = CENTRED descriptor head: convDb.bias is
    replaced by minus the mean of convDb's pre-bias output over the cells of a calibration frame (an LSUV-style data-dependent init;
    the vector for seed 7 is committed as data/sp_desc_center_seed7.npy, written by tools/gen_desc_center.py).  A random conv stack fed
    all-positive ReLU maps gives descriptors that share one large common component (cosine 0.83 +- 0.08 between ANY two keypoints of
    a frame), which no trained SuperPoint does and which LightGlue cannot match through; centred, random pairs sit at 0.02 +- 0.39 and
    corresponding keypoints of shifted frames at > 0.99.  The detector branch is untouched: keypoints and scores are those of the plain law."""
    rng = np.random.default_rng(seed)
    man, n = sp_manifest()
    blob = np.empty(n, np.float32)
    for name, off, shape in man:
        cnt = int(np.prod(shape))
        if name.endswith(".weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            w = rng.standard_normal(cnt).astype(np.float32) * np.float32(np.sqrt(2.0 / fan_in))
            if name.startswith("convPb"):
                w *= np.float32(logit_gain)
            blob[off:off + cnt] = w
        else:
            b = rng.uniform(-0.05, 0.05, cnt).astype(np.float32)
            if name.startswith("convPb"):
                b[64] += np.float32(dustbin_bias)
            blob[off:off + cnt] = b
    if desc_center is not None:
        if isinstance(desc_center, str):
            import os
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"sp_desc_center_seed{seed}.npy")
            if desc_center != "auto" or not os.path.exists(path):
                raise ValueError(f"make_superpoint: no committed descriptor centre for seed {seed} ({path}); run tools/gen_desc_center.py --seed {seed}")
            desc_center = np.load(path)
        c = np.asarray(desc_center, np.float32).reshape(256)
        off = next(o for nm, o, _ in man if nm == "convDb.bias")
        blob[off:off + 256] = -c
    return blob


def make_lightglue(seed=11, proj_gain=6.0, calibrated=False):
    """Seeded LightGlue weights.  Default law (rounds 1-4, every committed fixture): N(0, 1/fan_in) Linears, ffn.3 x 0.5, final_proj x 6 --
    token norms grow 1 -> 22 over the 9 layers and the log-assignment matrix reaches |850|, where ONE fp32 ulp is 6e-5 in the log domain:
    any two fp32 evaluation orders of that graph differ by 1-3e-4 in the match scores (tests/tolerances.py, profiles/r04_weight_scale.md).

    `calibrated=True` (round 5): the same random draws with ffn.3 x 0.125, Wo / Wv x 0.25, final_proj x 3.9 -- token norms stay O(1) (1 -> 5),
    the log-assignment peaks at 52-58 on constructed 1024-keypoint sets over seeds 3 / 11 / 29 (the range trained LightGlue logits live
    in; 63 on the bench's SuperPoint descriptors), ~1000 of 1024 constructed correspondences are found, and independent fp32
    evaluations (oracle, torch modules, graph execution) agree to ~6e-6.  north_star's 1e-4 bar is tested on this set."""
    rng = np.random.default_rng(seed)
    man, n = lg_manifest()
    blob = np.empty(n, np.float32)
    w2_gain, vo_gain = (0.125, 0.25) if calibrated else (0.5, 1.0)
    if calibrated:
        proj_gain = 3.9 if proj_gain == 6.0 else proj_gain
    for name, off, shape in man:
        cnt = int(np.prod(shape))
        leaf = name.split(".")[-1]
        if leaf == "Wr":
            v = rng.standard_normal(cnt) * 3.0
        elif leaf == "ln_g":
            v = 1.0 + 0.1 * rng.standard_normal(cnt)
        elif leaf in ("ln_b",) or leaf.startswith("b"):
            v = rng.uniform(-0.05, 0.05, cnt)
        elif leaf == "w":  # matchability
            v = rng.standard_normal(cnt) / np.sqrt(256.0)
        else:
            fan_in = shape[-1]
            g = 1.0
            if leaf == "W2":
                g = w2_gain  # keep the residual stream tame over 18 blocks
            if leaf in ("Wo", "Wv"):
                g = vo_gain
            if name.startswith("final_proj"):
                g = proj_gain
            v = rng.standard_normal(cnt) * (g / np.sqrt(fan_in))
        blob[off:off + cnt] = v.astype(np.float32)
    if name.endswith("matchability.b"):
        blob[-1] = np.float32(2.0)
    return blob


_MAGIC = b"RFEW"

# Graph hyper-parameters (include/rover_fe.h: rfe_hparams).  The reference's C++ never names them: they are constants inside its two
# .onnx files (K is read from the output tensor's shape, src/Extractors/superpoint_onnx.cc:169-181; the match filter is applied in the
# graph, src/Matchers/lightglue_onnx.cpp:404-409).  onnx_weights.py reads them from a graph; the RFEW v2 header carries them to
# rfe_load_weights.  Defaults = the published LightGlue-style export settings.
SP_HPARAMS = {"max_keypoints": 1024, "detection_threshold": 0.0005, "nms_radius": 4, "remove_borders": 4, "topk_always": 0}
LG_HPARAMS = {"layers": LG_LAYERS, "heads": 4, "filter_threshold": 0.1}


def pack_hparams(kind, hp=None):
    d = dict(SP_HPARAMS if kind == 1 else LG_HPARAMS)
    d.update(hp or {})
    if kind == 1:
        return struct.pack("<ifiii", int(d["max_keypoints"]), float(d["detection_threshold"]), int(d["nms_radius"]), int(d["remove_borders"]), int(d["topk_always"]))
    return struct.pack("<iif", int(d["layers"]), int(d["heads"]), float(d["filter_threshold"]))


def unpack_hparams(kind, blk):
    if kind == 1:
        k, t, r, b, a = struct.unpack_from("<ifiii", blk)
        return {"max_keypoints": k, "detection_threshold": t, "nms_radius": r, "remove_borders": b, "topk_always": a}
    l, h, t = struct.unpack_from("<iif", blk)
    return {"layers": l, "heads": h, "filter_threshold": t}


def save(path, blob, kind, hparams=None, version=2):
    """kind: 1 = SuperPoint, 2 = LightGlue.  version 2 (default) carries the graph hyper-parameters (`hparams`: dict over the keys
    of SP_HPARAMS / LG_HPARAMS, missing keys = defaults); version 1 is the bare container of earlier rounds (still loadable)."""
    blob = np.ascontiguousarray(blob, np.float32)
    with open(path, "wb") as f:
        f.write(_MAGIC + struct.pack("<IIQ", version, kind, blob.size))
        if version == 2:
            hb = pack_hparams(kind, hparams)
            f.write(struct.pack("<I", len(hb)) + hb)
        else:
            assert version == 1 and not hparams, "hyper-parameters need an RFEW v2 container"
        f.write(blob.tobytes())


def load(path, with_hparams=False):
    with open(path, "rb") as f:
        head = f.read(20)
        assert head[:4] == _MAGIC, "not an RFEW file"
        ver, kind, cnt = struct.unpack("<IIQ", head[4:])
        assert ver in (1, 2), f"unknown RFEW version {ver}"
        hp = dict(SP_HPARAMS if kind == 1 else LG_HPARAMS)
        if ver == 2:
            (hb,) = struct.unpack("<I", f.read(4))
            hp = unpack_hparams(kind, f.read(hb))
        blob = np.frombuffer(f.read(cnt * 4), np.float32).copy()
    assert blob.size == cnt
    return (blob, kind, hp) if with_hparams else (blob, kind)
