#!/usr/bin/env python3
"""Generate tests/golden/*.npz from an INDEPENDENT implementation (HuggingFace `transformers`
SuperPoint / LightGlue modelling code, architecture only) loaded with this repo's seeded synthetic
weights.  Runs in the build container only (needs `transformers`); the .npz fixtures are data
(inputs + expected outputs) and travel with the repo.

This is NOT the reference (the reference runs two ONNX blobs that are missing from its checkout,
.MISSING_LARGE_BLOBS:4-5) -- it pins the oracle's restatement of the published architectures
against a second, unrelated code base.  Differences handled here:
  * HF SuperPoint returns float keypoints relative to (W,H); we undo that.
  * HF threshold / NMS radius are configured to the LightGlue-ONNX export values (0.0005, 4);
    top-k is disabled (max_keypoints=-1) so the fixture holds the full candidate list in
    row-major order, and the 4-px border is applied here (HF's own border filter misses the
    right/bottom edges).
  * HF LightGlue early-stop / pruning are disabled by driving the layers directly.

usage: python tools/gen_golden.py            (writes tests/golden/sp_{a,b,c}.npz, lg_{a,b}.npz: small cases)
       python tools/gen_golden.py fullsize   (writes sp_{d,e}.npz, lg_{c,d}.npz: the sizes bench.py runs, top-k path, ragged pair)
       python tools/gen_golden.py oddsize    (writes sp_{f,g}.npz: 376 x 1241 and 101 x 151, sizes that are not multiples of 8)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rover_slam_amd as R  # noqa: E402
from rover_slam_amd import weights as Wt  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def hf_superpoint(blob, kmax):
    from transformers import SuperPointConfig, SuperPointForKeypointDetection
    cfg = SuperPointConfig(keypoint_threshold=0.0005, max_keypoints=kmax, nms_radius=4, border_removal_distance=4)
    m = SuperPointForKeypointDetection(cfg).eval()
    man, _ = Wt.sp_manifest()
    t = {name: torch.from_numpy(blob[off:off + int(np.prod(shape))].reshape(shape).copy()) for name, off, shape in man}
    enc = m.encoder.conv_blocks
    pairs = [(enc[0].conv_a, "conv1a"), (enc[0].conv_b, "conv1b"), (enc[1].conv_a, "conv2a"), (enc[1].conv_b, "conv2b"),
             (enc[2].conv_a, "conv3a"), (enc[2].conv_b, "conv3b"), (enc[3].conv_a, "conv4a"), (enc[3].conv_b, "conv4b"),
             (m.keypoint_decoder.conv_score_a, "convPa"), (m.keypoint_decoder.conv_score_b, "convPb"),
             (m.descriptor_decoder.conv_descriptor_a, "convDa"), (m.descriptor_decoder.conv_descriptor_b, "convDb")]
    with torch.no_grad():
        for mod, name in pairs:
            mod.weight.copy_(t[name + ".weight"])
            mod.bias.copy_(t[name + ".bias"])
    return m


def run_hf_superpoint(m, img_u8):
    H, W = img_u8.shape
    x = torch.from_numpy(img_u8.astype(np.float32) * np.float32(1.0 / 255.0))[None, None].repeat(1, 3, 1, 1)
    with torch.no_grad():
        out = m(pixel_values=x)
        feat = m.encoder(x[:, :1])[0]
        sc = m.keypoint_decoder.relu(m.keypoint_decoder.conv_score_a(feat))
        sc = m.keypoint_decoder.conv_score_b(sc)
        sc = torch.softmax(sc, 1)[:, :-1]
        b, _, h, w = sc.shape
        sc = sc.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
    n = int(out.mask[0].sum())
    kxy = (out.keypoints[0, :n] * torch.tensor([W, H])).round().to(torch.int32).numpy()
    score, desc = out.scores[0, :n].numpy(), out.descriptors[0, :n].numpy()
    # HF passes (height*8, width*8) of an already full-resolution map to its border filter, so the
    # right/bottom borders are never removed there; apply the published 4-px border on all sides.
    keep = (kxy[:, 0] >= 4) & (kxy[:, 0] < W - 4) & (kxy[:, 1] >= 4) & (kxy[:, 1] < H - 4)
    kxy, score, desc = kxy[keep], score[keep], desc[keep]
    return dict(n=int(keep.sum()), kxy=kxy, score=score, desc=desc, scoremap=sc[0].numpy())


def run_hf_superpoint_topk(m, img_u8, kmax):
    """Top-k path (count > Kmax, which every bench frame takes): HF's own functions in HF's own order -- pixel scores +
    simple_nms, threshold, border removal, torch.topk, descriptor sampling -- driven one by one so that
    remove_keypoints_from_borders gets the true score-map size (inside the model it is called with (8H, 8W) and never removes
    the right / bottom border, which would change WHICH keypoints make the top k).  The score map is 8*(H/8) x 8*(W/8), the
    frame the published SuperPoint (and the LightGlue-ONNX export the reference loads) blanks the 4-px border on; it equals the
    image when H and W are multiples of 8."""
    from transformers.models.superpoint import modeling_superpoint as MS
    H, W = img_u8.shape
    x = torch.from_numpy(img_u8.astype(np.float32) * np.float32(1.0 / 255.0))[None, None]
    with torch.no_grad():
        feat = m.encoder(x)[0]
        dec = m.keypoint_decoder
        scores = dec._get_pixel_scores(feat)
        kp = torch.nonzero(scores[0] > dec.keypoint_threshold)
        sc = scores[0][tuple(kp.t())]
        assert scores.shape[1:] == (H // 8 * 8, W // 8 * 8)
        kp, sc = MS.remove_keypoints_from_borders(kp, sc, dec.border_removal_distance, scores.shape[1], scores.shape[2])
        ncand = int(kp.shape[0])
        kp, sc = MS.top_k_keypoints(kp, sc, kmax)
        kxy = torch.flip(kp, [1]).to(sc.dtype)
        desc = m.descriptor_decoder(feat, kxy[None])
        assert desc.shape == (kxy.shape[0], 256)
    return dict(n=int(kxy.shape[0]), candidates=ncand, kxy=kxy.round().to(torch.int32).numpy(), score=sc.numpy(), desc=desc.numpy())


def hf_lightglue_modules(blob):
    from transformers import LightGlueConfig
    from transformers.models.lightglue import modeling_lightglue as ML
    cfg = LightGlueConfig(depth_confidence=-1.0, width_confidence=-1.0, filter_threshold=0.1)
    cfg._attn_implementation = "eager"
    man, _ = Wt.lg_manifest()
    t = {name: torch.from_numpy(blob[off:off + int(np.prod(shape))].reshape(shape).copy()) for name, off, shape in man}
    pos = ML.LightGluePositionalEncoder(cfg).eval()
    layers = [ML.LightGlueTransformerLayer(cfg, i).eval() for i in range(9)]
    assign = ML.LightGlueMatchAssignmentLayer(cfg).eval()
    with torch.no_grad():
        pos.projector.weight.copy_(t["posenc.Wr"])
        for l, L in enumerate(layers):
            p = f"layers.{l}."
            sa, ca = L.self_attention, L.cross_attention
            wqkv, bqkv = t[p + "self.Wqkv"], t[p + "self.bqkv"]
            sa.q_proj.weight.copy_(wqkv[0:256]); sa.q_proj.bias.copy_(bqkv[0:256])
            sa.k_proj.weight.copy_(wqkv[256:512]); sa.k_proj.bias.copy_(bqkv[256:512])
            sa.v_proj.weight.copy_(wqkv[512:768]); sa.v_proj.bias.copy_(bqkv[512:768])
            sa.o_proj.weight.copy_(t[p + "self.Wo"]); sa.o_proj.bias.copy_(t[p + "self.bo"])
            ca.q_proj.weight.copy_(t[p + "cross.Wqk"]); ca.q_proj.bias.copy_(t[p + "cross.bqk"])
            ca.k_proj.weight.copy_(t[p + "cross.Wqk"]); ca.k_proj.bias.copy_(t[p + "cross.bqk"])
            ca.v_proj.weight.copy_(t[p + "cross.Wv"]); ca.v_proj.bias.copy_(t[p + "cross.bv"])
            ca.o_proj.weight.copy_(t[p + "cross.Wo"]); ca.o_proj.bias.copy_(t[p + "cross.bo"])
            for mlp, tag in ((L.self_mlp, "self"), (L.cross_mlp, "cross")):
                mlp.fc1.weight.copy_(t[p + tag + ".W1"]); mlp.fc1.bias.copy_(t[p + tag + ".b1"])
                mlp.layer_norm.weight.copy_(t[p + tag + ".ln_g"]); mlp.layer_norm.bias.copy_(t[p + tag + ".ln_b"])
                mlp.fc2.weight.copy_(t[p + tag + ".W2"]); mlp.fc2.bias.copy_(t[p + tag + ".b2"])
        assign.final_projection.weight.copy_(t["final_proj.W"]); assign.final_projection.bias.copy_(t["final_proj.b"])
        assign.matchability.weight.copy_(t["matchability.w"][None]); assign.matchability.bias.copy_(t["matchability.b"])
    return ML, pos, layers, assign


def run_hf_lightglue(mods, k0n, k1n, d0, d1, thr=0.1):
    """M != N: both sides are padded to max(M, N) and HF's own masking (additive attention mask in the layers, 0/1 mask
    in the assignment) hides the padding."""
    ML, pos, layers, assign = mods
    M, N = len(k0n), len(k1n)
    L = max(M, N)
    padk = lambda k: np.concatenate([k, np.zeros((L - len(k), 2), np.float32)])
    padd = lambda d: np.concatenate([d, np.zeros((L - len(d), 256), np.float32)])
    kp = torch.from_numpy(np.stack([padk(k0n), padk(k1n)]))       # [2, L, 2]
    x = torch.from_numpy(np.stack([padd(d0), padd(d1)]))          # [2, L, 256]
    amask = mask01 = None
    if M != N:
        mask01 = torch.zeros(2, L)
        mask01[0, :M] = 1; mask01[1, :N] = 1
        amask = ((1.0 - mask01) * torch.finfo(torch.float32).min)[:, None, None, :]
    with torch.no_grad():
        enc = pos(kp)[0]
        for Lyr in layers:
            x = Lyr(x, enc, attention_mask=amask)[0]
        scores = assign(x, mask01)                     # [1, L+1, L+1]
        matches, mscores = ML.get_matches_from_scores(scores, thr)
    m0 = matches[0].numpy()[:M]
    idx = np.nonzero((m0 >= 0) & (m0 < N))[0]
    pairs = np.stack([idx, m0[idx]], 1).astype(np.int32)
    return dict(x0=x[0, :M].numpy(), x1=x[1, :N].numpy(), scores=scores[0, :M, :N].numpy(), pairs=pairs,
                ms=mscores[0].numpy()[idx])


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    # ---- SuperPoint ----
    for tag, (H, W, kmax, seed, dust) in {"a": (64, 96, -1, 7, 0.0), "b": (120, 160, -1, 7, 0.0),
                                          "c": (96, 128, -1, 9, 6.0)}.items():
        blob = Wt.make_superpoint(seed=seed, dustbin_bias=dust)
        frames, _ = R.synth.make_frames(1, H, W, seed=100 + H)
        hf = run_hf_superpoint(hf_superpoint(blob, kmax), frames[0])
        np.savez_compressed(os.path.join(GOLD, f"sp_{tag}.npz"), image=frames[0], seed=seed, dustbin_bias=dust,
                            kmax=kmax, n=hf["n"], kxy=hf["kxy"], score=hf["score"], desc=hf["desc"],
                            scoremap=hf["scoremap"].astype(np.float32))
        print(f"sp_{tag}: {H}x{W} n={hf['n']}")
    # ---- LightGlue ----
    lg = Wt.make_lightglue(seed=11)
    mods = hf_lightglue_modules(lg)
    rng = np.random.default_rng(5)
    for tag, n in {"a": 48, "b": 160}.items():
        # correlated descriptor sets: set 1 = permuted noisy copy of set 0 (gives real matches)
        d0 = rng.standard_normal((n, 256)).astype(np.float32)
        d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
        perm = rng.permutation(n)
        d1 = d0[perm] + 0.05 * rng.standard_normal((n, 256)).astype(np.float32)
        d1 = (d1 / np.linalg.norm(d1, axis=1, keepdims=True)).astype(np.float32)
        k0 = rng.uniform(-0.9, 0.9, (n, 2)).astype(np.float32)
        k1 = (k0[perm] + 0.02 * rng.standard_normal((n, 2))).astype(np.float32)
        hf = run_hf_lightglue(mods, k0, k1, d0, d1)
        np.savez_compressed(os.path.join(GOLD, f"lg_{tag}.npz"), seed=11, k0n=k0, k1n=k1, d0=d0, d1=d1, perm=perm,
                            x0=hf["x0"], x1=hf["x1"], scores=hf["scores"], pairs=hf["pairs"], ms=hf["ms"])
        print(f"lg_{tag}: n={n} matches={len(hf['pairs'])}")


def main_fullsize():
    """Fixtures at the sizes the benchmark runs: 480x640 and 480x752 SuperPoint through the top-k
    path (Kmax = 1024) on frame 0 of bench.py's own synthetic stream, LightGlue at M = N = 1024 and a ragged 700 x 1024 pair."""
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    wsp = Wt.make_superpoint(seed=7)
    m = hf_superpoint(wsp, 1024)
    frames, _ = R.synth.make_frames(2, 480, 640, seed=20240314)        # bench.py rank 0, frames 0 and 1
    wide, _ = R.synth.make_frames(1, 480, 752, seed=5)
    for tag, img in (("d", frames[0]), ("e", wide[0])):
        hf = run_hf_superpoint_topk(m, img, 1024)
        np.savez_compressed(os.path.join(GOLD, f"sp_{tag}.npz"), image=img, seed=7, dustbin_bias=0.0, kmax=1024, n=hf["n"],
                            candidates=hf["candidates"], kxy=hf["kxy"], score=hf["score"], desc=hf["desc"])
        print(f"sp_{tag}: {img.shape} n={hf['n']} of {hf['candidates']} candidates")
    wlg = Wt.make_lightglue(seed=11)
    mods = hf_lightglue_modules(wlg)
    # the recipe of lg_a / lg_b at full size: set 1 = permuted noisy copy of a random unit-vector set 0 (with random
    # weights two REAL frames yield only a handful of matches; this construction gives several hundred whose lists and
    # scores can be compared)
    rng = np.random.default_rng(2024)
    for tag, M, N in (("c", 1024, 1024), ("d", 700, 1024)):
        d0 = rng.standard_normal((1024, 256)).astype(np.float32)
        d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
        perm = rng.permutation(1024)
        d1 = d0[perm] + 0.01 * rng.standard_normal((1024, 256)).astype(np.float32)
        d1 = (d1 / np.linalg.norm(d1, axis=1, keepdims=True)).astype(np.float32)[:N]
        k0 = rng.uniform(-0.9, 0.9, (1024, 2)).astype(np.float32)
        k1 = (k0[perm] + 0.02 * rng.standard_normal((1024, 2))).astype(np.float32)[:N]
        k0, d0 = k0[:M], d0[:M]
        hf = run_hf_lightglue(mods, k0, k1, d0, d1)
        np.savez_compressed(os.path.join(GOLD, f"lg_{tag}.npz"), seed=11, k0n=k0, k1n=k1, d0=d0, d1=d1, perm=perm,
                            x0_rows4=hf["x0"][::4], x1_rows4=hf["x1"][::4], pairs=hf["pairs"], ms=hf["ms"])
        print(f"lg_{tag}: {M} x {N} matches={len(hf['pairs'])}")


def calibrated_case(tag):
    """inputs of the calibrated-weight fixtures lg_e / lg_f, regenerated from the seed by generator and tests alike (the fixtures hold outputs only)"""
    M, N, seed = {"e": (1024, 1024, 2025), "f": (700, 1024, 2026)}[tag]
    rng = np.random.default_rng(seed)
    d0 = rng.standard_normal((1024, 256)).astype(np.float32)
    d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    perm = rng.permutation(1024)
    d1 = d0[perm] + 0.01 * rng.standard_normal((1024, 256)).astype(np.float32)
    d1 = (d1 / np.linalg.norm(d1, axis=1, keepdims=True)).astype(np.float32)[:N]
    k0 = rng.uniform(-0.9, 0.9, (1024, 2)).astype(np.float32)
    k1 = (k0[perm] + 0.02 * rng.standard_normal((1024, 2))).astype(np.float32)[:N]
    return np.ascontiguousarray(k0[:M]), k1, np.ascontiguousarray(d0[:M]), d1, perm


def main_calibrated():
    """Round 5: the HF modules on the CALIBRATED LightGlue law (weights.make_lightglue(calibrated=True): logits in the range trained weights live in) at
    M = N = 1024 and ragged 700 x 1024 -- an independent implementation for north_star's 1e-4 bar (tests/test_oracle_golden.py, tests/test_gpu_calibrated.py)."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    mods = hf_lightglue_modules(Wt.make_lightglue(seed=11, calibrated=True))
    for tag in ("e", "f"):
        k0, k1, d0, d1, perm = calibrated_case(tag)
        hf = run_hf_lightglue(mods, k0, k1, d0, d1)
        np.savez_compressed(os.path.join(GOLD, f"lg_{tag}.npz"), seed=11, calibrated=1, perm=perm, x0_rows16=hf["x0"][::16], x1_rows16=hf["x1"][::16],
                            pairs=hf["pairs"], ms=hf["ms"])
        print(f"lg_{tag}: {len(k0)} x {len(k1)} matches={len(hf['pairs'])}")


def main_oddsize():
    """Image sizes that are not multiples of 8 (the reference graph has dynamic axes; KITTI is 1241 x 376): sp_f = 376 x 1241 through
    the top-k path, sp_g = 101 x 151 with every candidate kept.  The pools floor, the score map is 8*(H/8) x 8*(W/8)."""
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    wsp = Wt.make_superpoint(seed=7)
    for tag, (H, W), kmax, seed in (("f", (376, 1241), 1024, 31), ("g", (101, 151), 4096, 32)):
        img = R.synth.make_frames(1, H, W, seed=seed)[0][0]
        hf = run_hf_superpoint_topk(hf_superpoint(wsp, kmax), img, kmax)
        np.savez_compressed(os.path.join(GOLD, f"sp_{tag}.npz"), image=img, seed=7, dustbin_bias=0.0, kmax=kmax, n=hf["n"],
                            candidates=hf["candidates"], kxy=hf["kxy"], score=hf["score"], desc=hf["desc"])
        print(f"sp_{tag}: {img.shape} n={hf['n']} of {hf['candidates']} candidates")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize":
        main_fullsize()
    elif len(sys.argv) > 1 and sys.argv[1] == "oddsize":
        main_oddsize()
    elif len(sys.argv) > 1 and sys.argv[1] == "calibrated":
        main_calibrated()
    else:
        main()
