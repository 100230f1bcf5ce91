#!/usr/bin/env python3
"""bench.py -- frames/s of the SuperPoint + LightGlue front end on synthetic 640x480 grayscale frames.

Workload (BASELINE.json configs[3], per GPU): 33 device-resident u8 frames are extracted and frame i is
matched with frame i+1 (32 pairs) through rfe_extract_match_stream_dev; 32 frames are counted per
GPU per step (the 33rd is the one-frame overlap that makes the ranks independent).
Kmax = 1024, detection threshold 0.0005, match filter 0.1, seeded synthetic weights.

Multi-GPU: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI), frames sharded, no data-path
collective except ONE gather per step of the compact results (counts, keypoints, matches) to rank 0.
  * under torchrun (WORLD_SIZE set) this process is one rank;
  * `python bench.py --gpus N` with N > 1 and no WORLD_SIZE spawns the N ranks itself -- as child processes, before
    anything in this process touches the GPU -- and exits with their status;
  * `--gpus` that disagrees with WORLD_SIZE, or more ranks than visible GPUs (nccl), is an error, never a silent
    1-GPU number.
`--scaling weak` (default): 32 frames per GPU per step.  `--scaling strong`: 256 frames per step in all, 256/N per GPU.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, timed with HIP events
on the library's stream inside the timed region), `cpu_baseline` (the CPU oracle, rank 0, N=1 only, bounded sample),
`sustained` (>= 200 further steps: per-step spread, clocks) and, for N > 1, `ranks` / `rccl` (what the process group saw).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))   # tests/tolerances.py: the ONE statement of the LightGlue tolerances / borderline rule
# the host driver of this pool only supports dmabuf IPC; RCCL between processes fails without it (hipIpcGetMemHandle: invalid
# argument).  Exported by the image already -- set here too so that a torchrun launch from a clean environment works.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

H, W, KMAX = 480, 640, 1024
FRAMES_PER_GPU = 32
STRONG_TOTAL_FRAMES = 256   # BASELINE configs[3]: 256 frames over the node
PEAK_F32_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
PEAK_HBM_GBS = 8000.0

# SuperPoint 3x3 layers: (stage, Cin, Cout, downscale)
SP_CONV = {"conv1b": (64, 64, 1), "conv2a": (64, 64, 2), "conv2b": (64, 64, 2), "conv3a": (64, 128, 4),
           "conv3b": (128, 128, 4), "conv4a": (128, 128, 8), "conv4b": (128, 128, 8), "convPa": (128, 256, 8),
           "convDa": (128, 256, 8)}


def stage_flops(name, B, lens, launches=None):
    """ALGORITHMIC FLOPs of one launch of a profiled stage (DESIGN.md 'Work per unit')."""
    P = B - 1
    rows = 2 * P * KMAX
    if name in SP_CONV:
        cin, cout, ds = SP_CONV[name]
        return 2.0 * 9 * cin * cout * (H // ds) * (W // ds) * B
    if name == "conv1ab":   # conv1a (1->64) recomputed in LDS + conv1b (64->64): count both layers once
        return 2.0 * 9 * (1 * 64 + 64 * 64) * H * W * B
    if name == "lg_cross_qkv":
        return 2.0 * rows * 256 * 512
    if name == "convPb":
        return 2.0 * 256 * 65 * (H // 8) * (W // 8) * B
    if name == "convDb":
        return 2.0 * 256 * 256 * (H // 8) * (W // 8) * B
    # stream mode runs layer 0's self block once per FRAME (B sequences) instead of per pair side (2P sequences)
    rows_f = B * KMAX
    if name == "lg_qkv":            # 9 launches: 8 on 2P sequences + 1 on B frames
        return 2.0 * 256 * 768 * (8 * rows + rows_f) / 9
    if name == "lg_proj":
        if launches == 1:           # RFE_OPT_LG_FOLD_WO: the attention out-projections are folded into ffn.0 -> final_proj only
            return 2.0 * 256 * 256 * rows
        return 2.0 * 256 * 256 * (18 * rows + rows_f) / 19   # 9 self (one on frames) + 9 cross + final_proj
    if name == "lg_ffn1":           # 18 launches, one on frames
        return 2.0 * 512 * 512 * (17 * rows + rows_f) / 18
    if name == "lg_ffn2":
        return 2.0 * 512 * 256 * (17 * rows + rows_f) / 18
    if name == "lg_sim":
        return 2.0 * 256 * float(np.sum(lens[:-1].astype(np.float64) * lens[1:]))
    if name == "lg_attention":
        # average over the 18 launches (9 self, 9 cross): 4 heads * (QK^T + PV) * 64 dims
        a, b = lens[:-1].astype(np.float64), lens[1:].astype(np.float64)
        self_pairs = 4 * 4.0 * 64 * float(np.sum(a * a) + np.sum(b * b))       # one self launch on 2P sequences
        self_frames = 4 * 4.0 * 64 * float(np.sum(lens.astype(np.float64) ** 2))  # layer 0: once per frame
        cross_f = 4 * 4.0 * 64 * float(2 * np.sum(a * b))
        return (8 * self_pairs + self_frames + 9 * cross_f) / 18
    return None


STAGE_KERNEL = {"lg_attention": ("lg_attention_kernel (self + cross", "lg_attention_dma_kernel", "lg_attention_kernel"), "conv1ab": ("conv1ab_fused_kernel",)}


def pmc_traffic(stage):
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 PMC passes of this same
    bench command (profiles/rNN_pmc_traffic.json, made by tools/profile_round.sh + tools/rocpd_pmc.py):
    (2*FETCH_SIZE + WRITE_SIZE)*1024 as the MI355X guide prescribes.  None when no PMC pass covers it."""
    keys = STAGE_KERNEL.get(stage)
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")
        if not keys or not os.path.exists(path):
            continue
        kernels = json.load(open(path))["kernels"]
        for key in keys:        # rounds 2-4: one call-weighted entry for the attention's self (rotary) and cross kernels; round 5: one kernel, lg_attention_dma_kernel
            for name, v in kernels.items():
                if key in name:
                    return v["traffic_bytes"], f"profiles/{rnd}_pmc_traffic.json:" + name
    return None, None


def gpu_clocks(dev_index):
    """Current shader / memory clock from sysfs when the (unprivileged) process may read it; None otherwise."""
    out = {}
    try:
        cards = sorted(d for d in os.listdir("/sys/class/drm") if d.startswith("card") and d[4:].isdigit())
        card = cards[dev_index] if dev_index < len(cards) else cards[0]
        for key, fn in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")):
            with open(f"/sys/class/drm/{card}/device/{fn}") as f:
                for line in f:
                    if "*" in line:
                        out[key] = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
    except (OSError, ValueError, IndexError):
        pass
    if out.get("sclk_mhz", 1 << 30) < 500:      # some boxes expose only the idle level here (96 / 158 MHz under a 100 % busy chip): not a reading
        out = {}
    return out or None


def gpu_numa(dev_index):
    """NUMA node and local CPU list of the GPU from sysfs (None when unreadable): which host cores sit next to the device"""
    try:
        cards = sorted(d for d in os.listdir("/sys/class/drm") if d.startswith("card") and d[4:].isdigit())
        card = cards[dev_index] if dev_index < len(cards) else cards[0]
        out = {}
        for key in ("numa_node", "local_cpulist"):
            with open(f"/sys/class/drm/{card}/device/{key}") as f:
                out[key] = f.read().strip()
        out["process_affinity_cpus"] = len(os.sched_getaffinity(0))
        return out
    except (OSError, IndexError):
        return None


def cpu_model():
    """CPU model string of the host (core count AND model next to the CPU baseline)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def ort_reference_baseline(frames):
    """if onnxruntime AND the reference's two model files are available (RFE_ONNX_DIR; the reference
    checkout itself is never consulted), time the true reference arithmetic on the CPU execution provider.  Neither
    exists in the build image, so this returns None there (kept so that a site which has both gets kind = "reference")."""
    d = os.environ.get("RFE_ONNX_DIR")
    if not d:
        return None
    sp_path, lg_path = os.path.join(d, "superpoint.onnx"), os.path.join(d, "lightglue_sim.onnx")
    if not (os.path.exists(sp_path) and os.path.exists(lg_path)):
        return None
    try:
        import onnxruntime as ort
    except ImportError:
        return None
    so = ort.SessionOptions()
    so.intra_op_num_threads = len(os.sched_getaffinity(0))
    sp = ort.InferenceSession(sp_path, so, providers=["CPUExecutionProvider"])
    lg = ort.InferenceSession(lg_path, so, providers=["CPUExecutionProvider"])
    t0 = time.perf_counter()
    prev, nf = None, 0
    while nf < len(frames) and (nf < 4 or time.perf_counter() - t0 < 12.0):
        k, sc, de = sp.run(["keypoints", "scores", "descriptors"], {"image": (frames[nf].astype(np.float32) / 255.0)[None, None]})
        kn = ((k[0].astype(np.float32) - np.array([W / 2, H / 2], np.float32)) / (max(W, H) / 2)).astype(np.float32)
        if prev is not None:
            lg.run(["matches0", "mscores0"], {"kpts0": prev[0][None], "kpts1": kn[None], "desc0": prev[1][None], "desc1": de[0][None]})
        prev = (kn, de[0])
        nf += 1
    dt = time.perf_counter() - t0
    return {"value": round((nf - 1) / dt, 4), "unit": "frames/s", "cores": so.intra_op_num_threads, "cpu_model": cpu_model(), "kind": "reference",
            "sample": f"{nf} frames + {nf - 1} pairs through onnxruntime {ort.__version__} CPUExecutionProvider with the reference's "
                      f"superpoint.onnx / lightglue_sim.onnx from RFE_ONNX_DIR, {dt:.1f} s"}


def torch_cpu_baseline(frames, wsp, wlg, budget_s=10.0):
    """SURVEY 8(d) also asks for a torch-CPU leg: the same two networks as HuggingFace `transformers` modules (the
    independent implementation behind tests/golden, tools/gen_golden.py) on torch-CPU / oneDNN with all usable cores.
    A port like the oracle, only with a vendor-tuned convolution / GEMM back end -- closer to what ONNXRuntime-CPU does."""
    try:
        import torch
        from oracle import oracle as O
        from tools import gen_golden as G
        sp = G.hf_superpoint(wsp, KMAX)
        mods = G.hf_lightglue_modules(wlg)
    except Exception as e:   # transformers missing / incompatible: report why instead of failing the bench
        return {"value": None, "kind": "port", "error": f"{type(e).__name__}: {e}"[:200]}
    nthr = O.usable_cpus()
    torch.set_num_threads(nthr)
    t0 = time.perf_counter()
    prev, nf = None, 0
    while nf < len(frames) and (nf < 3 or time.perf_counter() - t0 < budget_s):
        r = G.run_hf_superpoint(sp, frames[nf])
        kn = ((r["kxy"].astype(np.float32) - np.array([W / 2, H / 2], np.float32)) / (max(W, H) / 2)).astype(np.float32)
        if prev is not None:
            m = min(len(kn), len(prev[0]))     # the HF layer stack is driven without masks: equal lengths
            G.run_hf_lightglue(mods, prev[0][:m], kn[:m], prev[1][:m], r["desc"][:m])
        prev = (kn, r["desc"])
        nf += 1
    dt = time.perf_counter() - t0
    return {"value": round((nf - 1) / dt, 4), "unit": "frames/s", "cores": nthr, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"{nf} frames + {nf - 1} pairs through HuggingFace transformers SuperPoint / LightGlue modules on torch-CPU "
                      f"({torch.__version__}, {nthr} threads), same synthetic weights and frames, {dt:.1f} s"}


def cpu_baseline(frames, wsp, wlg, gpu=None, tol=None, verify_budget_s=90.0):
    """The CPU oracle (a port, not the reference's ONNXRuntime path -- that cannot run here: no
    onnxruntime, no .onnx blobs) on a bounded sample of the bench frames (about 12 s of CPU work).  The oracle doubles as the checker of
    what the timed GPU loop produced: every frame / pair of the sample, and -- untimed, after the sample's clock has stopped -- the
    remaining pairs of the batch too (bounded by `verify_budget_s`), so that the match lists are compared on thousands of matches."""
    from oracle import oracle as O
    O.build()
    # bounded sample: frames are extracted and matched to their predecessor one by one until ~12 s of CPU work are spent
    t0 = time.perf_counter()
    prev, nf = None, 0
    sp_ok, lg_ok, lg_identical = True, True, True
    ms_dev, n_matches, one_sided = 0.0, 0, 0
    alt = gpu.get("alt") if gpu is not None else None      # match lists of the same pairs from variants.fp16x2 (RFE_OPT_LG_FP16X2 on): same oracle, same rule
    alt_stat = {"ok": True, "identical": True, "dev": 0.0, "matches": 0, "one_sided": 0}
    from tolerances import LG_SCORE_TOL, lists_agree_borderline
    LG_TOL = LG_SCORE_TOL if tol is None else tol

    def check_pair(idx, lg):
        """the tests' own rule (tests/tolerances.py): lists equal up to borderline flips (score within tol of the 0.1 filter, or
        the two best candidates of a row / column closer than 2 tol), common scores within LG_SCORE_TOL"""
        nonlocal lg_ok, lg_identical, ms_dev, n_matches, one_sided
        Sg = int(gpu["S"][idx])
        ok, dev, only = lists_agree_borderline(gpu["pairs"][idx, :Sg], gpu["ms"][idx, :Sg], lg["pairs"], lg["ms"], lg["scores"], KMAX, tol=LG_TOL)
        lg_ok &= bool(ok and dev < LG_TOL)
        lg_identical &= bool(only == 0 and Sg == lg["S"])
        ms_dev, n_matches, one_sided = max(ms_dev, float(dev)), n_matches + Sg, one_sided + only
        if alt is not None:
            Sa = int(alt["S"][idx])
            ok2, dev2, only2 = lists_agree_borderline(alt["pairs"][idx, :Sa], alt["ms"][idx, :Sa], lg["pairs"], lg["ms"], lg["scores"], KMAX, tol=LG_TOL)
            alt_stat["ok"] &= bool(ok2 and dev2 < LG_TOL)
            alt_stat["identical"] &= bool(only2 == 0 and Sa == lg["S"])
            alt_stat["dev"], alt_stat["matches"], alt_stat["one_sided"] = max(alt_stat["dev"], float(dev2)), alt_stat["matches"] + Sa, alt_stat["one_sided"] + only2

    while nf < len(frames) and (nf < 4 or time.perf_counter() - t0 < 12.0):
        cur = O.superpoint(wsp, frames[nf], kmax=KMAX)
        lg = None
        if prev is not None:
            lg = O.lightglue(wlg, O.normalize_keypoints(prev["kxy"][:prev["n"]].astype(np.float32), H, W),
                             O.normalize_keypoints(cur["kxy"][:cur["n"]].astype(np.float32), H, W), prev["desc"][:prev["n"]], cur["desc"][:cur["n"]],
                             debug=gpu is not None)     # debug: also the log-assignment matrix, for the borderline rule of the check
        if gpu is not None:   # the oracle doubles as the checker of what the timed loop produced (not timed here: numpy compares are cheap)
            sp_ok &= bool(gpu["n"][nf] == cur["n"] and np.array_equal(gpu["kxy"][nf], cur["kxy"]) and np.array_equal(gpu["score"][nf], cur["score"])
                          and np.array_equal(gpu["desc"][nf], cur["desc"]))
            if lg is not None:
                check_pair(nf - 1, lg)
        prev = cur
        nf += 1
    dt = time.perf_counter() - t0
    verified = None
    if gpu is not None:
        # untimed: the REST of the batch (the sample's clock has stopped) -- every remaining frame and pair while the budget lasts, and in any
        # case the last pair (the batch edges are where indexing slips show)
        extra_f, extra_p, t1 = 0, 0, time.perf_counter()
        k = nf
        while k < len(frames):
            if time.perf_counter() - t1 > verify_budget_s and k < len(frames) - 1:
                prev, k = O.superpoint(wsp, frames[-2], kmax=KMAX), len(frames) - 1     # budget spent: jump to the last pair
                continue
            cur = O.superpoint(wsp, frames[k], kmax=KMAX)
            sp_ok &= bool(gpu["n"][k] == cur["n"] and np.array_equal(gpu["kxy"][k], cur["kxy"]) and np.array_equal(gpu["score"][k], cur["score"])
                          and np.array_equal(gpu["desc"][k], cur["desc"]))
            lg = O.lightglue(wlg, O.normalize_keypoints(prev["kxy"][:prev["n"]].astype(np.float32), H, W),
                             O.normalize_keypoints(cur["kxy"][:cur["n"]].astype(np.float32), H, W), prev["desc"][:prev["n"]], cur["desc"][:cur["n"]], debug=True)
            check_pair(k - 1, lg)
            prev, k, extra_f, extra_p = cur, k + 1, extra_f + 1, extra_p + 1
        verified = {"ok": bool(sp_ok and lg_ok), "frames": nf + extra_f, "pairs": nf - 1 + extra_p, "pairs_in_batch": len(gpu["S"]), "superpoint_bit_exact": sp_ok,
                    "match_lists_agree": lg_ok, "match_lists_identical": lg_identical, "matches_compared": n_matches,
                    "matches_per_pair_mean": round(n_matches / max(nf - 1 + extra_p, 1), 1),
                    "one_sided_borderline_matches": one_sided, "match_score_max_dev": ms_dev, "match_score_tolerance": LG_TOL,
                    "lg_fold_wo": gpu.get("fold"), "untimed_seconds_beyond_the_sample": round(time.perf_counter() - t1, 1),
                    "rule": "tests/tolerances.py lists_agree_borderline: a match only one side reports must sit within tol of the 0.1 filter or "
                            "on a row / column whose two best probabilities are closer than 2 tol; common scores within tol",
                    "tolerance_note": ("north_star's 1e-4 on the calibrated LightGlue weight set (tests/tolerances.py LG_SCORE_TOL_CALIBRATED: log-assignment peaks at "
                                       "52-63, independent fp32 evaluations agree to ~6e-6)" if LG_TOL <= 1e-4 else
                                       "stated fp32 tolerance of LightGlue match scores at K = 1024 on the ill-conditioned seeded weight set (tests/tolerances.py, "
                                       "profiles/r02_lg_tolerance.md: any two fp32 evaluations of the graph differ by 1-3e-4, oracle vs float64 2.6e-4)")}
        if alt is not None:
            verified["fp16x2_variant"] = {"ok": alt_stat["ok"], "match_lists_identical": alt_stat["identical"], "matches_compared": alt_stat["matches"],
                                          "one_sided_borderline_matches": alt_stat["one_sided"], "match_score_max_dev": alt_stat["dev"],
                                          "note": "the match lists variants.fp16x2 produced for the same pairs, against the same oracle results under the same rule "
                                                  "(SuperPoint is not affected by the option)"}
        if not verified["ok"]:
            print(f"bench.py: GPU results of the timed loop differ from the oracle: {verified}", file=sys.stderr)
    return {"value": round((nf - 1) / dt, 4), "unit": "frames/s", "cores": O.threads(), "cpu_model": cpu_model(), "kind": "port",
            "verified_against_gpu": verified,
            "sample": f"{nf} frames 640x480 extracted + {nf - 1} consecutive pairs matched (K<=1024) by oracle/rfe_oracle.c, "
                      f"OpenMP on {O.threads()} threads (= the CPUs this process may use: {os.cpu_count()} logical CPUs, affinity and "
                      f"cgroup quota applied), {dt:.1f} s; {nf - 1} frames counted"}


def oracle_mismatches(out):
    """Every place of a bench line where an auxiliary run was CHECKED AGAINST THE ORACLE (or failed inside the library) and lost: the drop-in classes
    (`latency.dropin`, `latency.dropin_host_graph`), the one-pair entry points (`latency.resident`, `latency.resident_fp16x2`), `variants.fp16x2`.  Each
    is a correctness failure of a shipped path -> out["invalid_aux"], exit status 5.  Timings never enter here (they live in out["perf_notes"]);
    infrastructure failures (a missing driver binary, OOM) keep their `error` string and exit 0.  Pure function of the line (tests/test_bench_exit.py)."""
    bad = []
    for k, v in sorted((out.get("latency") or {}).items()):
        if not isinstance(v, dict):
            continue
        ver = v.get("verified_against_oracle")
        if isinstance(ver, dict) and ver.get("ok") is False:
            bad.append(f"latency.{k}: results differ from the oracle ({', '.join(f'{a}={b}' for a, b in ver.items() if a != 'ok')})"[:400])
        err = v.get("error")
        if isinstance(err, str) and err.startswith("RfeError"):
            bad.append(f"latency.{k}: the library refused or failed the call: {err}"[:400])
    for k, v in sorted((out.get("variants") or {}).items()):
        if k == "error" and isinstance(v, str) and v.startswith("RfeError"):
            bad.append(f"variants: the library refused or failed a call: {v}"[:400])
        if not isinstance(v, dict):
            continue
        ver = v.get("verified_against_oracle")
        if isinstance(ver, dict) and ver.get("ok") is False:
            bad.append(f"variants.{k}: results differ from the oracle ({', '.join(f'{a}={b}' for a, b in ver.items() if a != 'ok')})"[:400])
        err = v.get("error")
        if isinstance(err, str) and err.startswith("RfeError"):
            bad.append(f"variants.{k}: the library refused or failed the call: {err}"[:400])
    return bad


def exit_status(out, aux_mismatch):
    """The line's verdict -> (exit status, line with `invalid` / `invalid_aux` filled in).  4: the timed loop's own outputs differ from the oracle (no
    headline value); 5: an auxiliary run's RESULTS are wrong; 0 otherwise.  No timing decides anything here."""
    rc = 0
    ver = (out.get("cpu_baseline_port") or out.get("cpu_baseline") or {}).get("verified_against_gpu")
    if ver is not None and not ver["ok"]:
        # a fast kernel whose results differ from the oracle's is not a result: no headline number, non-zero exit status
        out["invalid"] = "outputs of the timed loop differ from the CPU oracle beyond the stated tolerance (cpu_baseline.verified_against_gpu)"
        out["value_unverified"], out["value"] = out.get("value"), None
        rc = 4
    aux = list(aux_mismatch) + oracle_mismatches(out)
    if aux:
        out["invalid_aux"] = aux
        rc = rc or 5
    return rc


class AuxMismatch(RuntimeError):
    """an auxiliary run (variants / PCIe pipeline / pool) produced RESULTS that differ from the resident path: a correctness failure, reported
    as out["invalid_aux"] with a non-zero exit status -- unlike infrastructure failures (OOM, missing driver), which only leave an `error` string"""


def stereo_frames(synth, T=8):
    """T synthetic 752x480 stereo pairs (EuRoC size): one scene, the right view shifted by a 24-px disparity, the camera moving 8 px
    per frame.  -> (lefts, rights): lists of u8 [480,752]"""
    Hs, Ws = 480, 752
    rng = np.random.default_rng(5)
    scene = synth.make_scene(rng, Hs, Ws + 64 + 8 * T, margin=0)
    lefts, rights = [], []
    for t in range(T):
        disp = 24
        x0 = 8 * t
        lefts.append(np.clip(scene[:, x0:x0 + Ws] + rng.integers(0, 8, (Hs, Ws)), 0, 255).astype(np.uint8))
        rights.append(np.clip(scene[:, x0 + disp:x0 + disp + Ws] + rng.integers(0, 8, (Hs, Ws)), 0, 255).astype(np.uint8))
    return lefts, rights


def run_stereo_stream(ctx, capi, synth, torch, dev, kmax, steps, warmup, with_stages=True):
    """BASELINE configs[4]: a 752x480 stereo stream through ONE device-resident entry point per stereo frame,
    rfe_stereo_frame_dev: both views through SuperPoint as a batch of 2 (src/Frame.cc:142-147), Frame::ComputeStereoMatches
    (src/Frame.cc:1159-1446) and one LightGlue match of the left view against the previous left view
    (SPmatcher.cc:1050-1080).  Nothing crosses PCIe inside the loop.  Latency figure, not the metric's workload."""
    Hs, Ws, K = 480, 752, kmax
    T = 8                                                     # distinct stereo frames, cycled
    lefts, rights = stereo_frames(synth, T)
    imgs = torch.from_numpy(np.stack([np.stack([l, r]) for l, r in zip(lefts, rights)])).to(dev)      # [T,2,H,W]
    st = capi.StereoStream(ctx, Hs, Ws, K, mb=0.11, mbf=0.11 * 435.0)

    def step(t):
        im = imgs[t % T]
        st.push(im[0].data_ptr(), im[1].data_ptr(), Ws)

    for t in range(warmup + 1):
        step(t)
    torch.cuda.synchronize(dev)
    prof, dt_prof = {}, None
    if with_stages:
        ctx.profile(True); ctx.profile_reset()
        t0 = time.perf_counter()
        for t in range(steps):
            step(warmup + 1 + t)
        torch.cuda.synchronize(dev)
        dt_prof = time.perf_counter() - t0
        prof = ctx.profile_read(); ctx.profile(False)
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()                                   # headline: the same loop without the per-stage events
    for t in range(steps):
        step(warmup + 1 + steps + t)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    res = st.results()
    st.close()
    out = {"metric": "BASELINE configs[4] stereo 752x480 stream, latency run", "value": round(steps / dt, 2),
           "unit": "stereo frames/s", "ms_per_step": round(dt / steps * 1e3, 4), "n_gpus": 1,
           "steps": steps, "warmup": warmup, "kmax": K, "left_keypoints": int(res["n"][0]),
           "stereo_matches": int((res["u_right"][:int(res["n"][0])] >= 0).sum()), "temporal_matches": int(res["S"]),
           "entry_point": "rfe_stereo_frame_dev (no host synchronisation inside the loop)"}
    if with_stages:
        out["ms_per_step_with_stage_events"] = round(dt_prof / steps * 1e3, 4)
        out["stages_ms_per_step"] = {k: round(v[0] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
    return out


def bench_stereo_stream(args, ctx, capi, synth, torch, dev, rank):
    out = run_stereo_stream(ctx, capi, synth, torch, dev, args.kmax, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(out))


def oracle_check_pair(pair_np, wsp, wlg, res, tol):
    """One 640x480 pair as the device-resident one-pair entry points left it (`res`: n, kxy, score, desc of both frames, S / pairs / ms of the pair)
    against the CPU oracle: SuperPoint bit for bit, the match list under tests/tolerances.py's rule.  Checker only (untimed)."""
    from oracle import oracle as O
    from tolerances import lists_agree_borderline
    O.build()
    o = [O.superpoint(wsp, pair_np[i], kmax=KMAX) for i in range(2)]
    sp = all(bool(res["n"][i] == o[i]["n"] and np.array_equal(res["kxy"][i], o[i]["kxy"]) and np.array_equal(res["score"][i], o[i]["score"])
                  and np.array_equal(res["desc"][i], o[i]["desc"])) for i in range(2))
    lg = O.lightglue(wlg, O.normalize_keypoints(o[0]["kxy"][:o[0]["n"]].astype(np.float32), H, W), O.normalize_keypoints(o[1]["kxy"][:o[1]["n"]].astype(np.float32), H, W),
                     o[0]["desc"][:o[0]["n"]], o[1]["desc"][:o[1]["n"]], debug=True)
    Sg = int(res["S"][0])
    ok, dev, only = lists_agree_borderline(res["pairs"][0, :Sg], res["ms"][0, :Sg], lg["pairs"], lg["ms"], lg["scores"], KMAX, tol=tol)
    return {"ok": bool(sp and ok and dev < tol), "superpoint_bit_exact": bool(sp), "match_list_agrees": bool(ok), "matches": Sg, "matches_oracle": int(lg["S"]),
            "one_sided_borderline": int(only), "match_score_max_dev": float(dev), "match_score_tolerance": tol}


def latency_resident(ctx, capi, synth, sharding, torch, dev, frames, steps=200, warmup=20, check=None):
    """`latency.resident` of the bench line: BASELINE configs[1] / [2] / [4] -- the shapes the reference itself runs (batch 1,
    src/Extractors/superpoint_onnx.cc:100, src/Matchers/lightglue_onnx.cpp:168-172) -- device-resident, `steps` calls each, no
    per-stage events, wall clock between two device synchronisations.  `frames`: the first two frames of the bench stream (device)."""
    pack = sharding.ResultPack(2, KMAX, dev)
    a = (pack.n.data_ptr(), pack.kxy.data_ptr(), pack.score.data_ptr(), pack.desc.data_ptr())

    def c2():
        ctx._chk(capi.lib.rfe_extract_u8_dev(ctx.h, frames.data_ptr(), H, W, W, 1, KMAX, 0.0005, *a))

    def c3():
        ctx._chk(capi.lib.rfe_extract_match_stream_dev(ctx.h, frames.data_ptr(), H, W, W, 2, KMAX, 0.0005, 0.1, *a, pack.S.data_ptr(),
                                                       pack.pairs.data_ptr(), pack.ms.data_ptr()))
    out = {}
    for name, fn, what in (("c2", c2, "configs[1]: SuperPoint, one 640x480 frame, rfe_extract_u8_dev"),
                           ("c3", c3, "configs[2]: one 640x480 pair = 2 extractions + 1 LightGlue match (K <= 1024), rfe_extract_match_stream_dev B = 2")):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize(dev)
        out[name] = {"ms": round((time.perf_counter() - t0) / steps * 1e3, 4), "steps": steps, "what": what}
    out["c2"]["keypoints"] = int(pack.n[0].item())
    out["c3"]["matches"] = int(pack.S[0].item())
    if check is not None:     # check = (pair_np, wsp, wlg, tol): the one-pair kernels' own results (c3's last call) against the oracle
        res = {k: getattr(pack, k).cpu().numpy() for k in ("n", "kxy", "score", "desc", "S", "pairs", "ms")}
        out["verified_against_oracle"] = oracle_check_pair(*check[:3], res, check[3])
    r5 = run_stereo_stream(ctx, capi, synth, torch, dev, KMAX, steps, warmup, with_stages=False)
    out["c5"] = {"ms": r5["ms_per_step"], "steps": steps, "stereo_matches": r5["stereo_matches"], "temporal_matches": r5["temporal_matches"],
                 "what": "configs[4]: one 752x480 stereo frame = batch-of-2 extraction + ComputeStereoMatches + 1 LightGlue match against the "
                         "previous left view, rfe_stereo_frame_dev"}
    return out


def latency_dropin(synth, wsp, wlg, pair_np, steps=50, warmup=5, check=True, tol=None, host_graph=False):
    """`latency.dropin`: the same three configurations through the C++ drop-in classes with HOST pointers in and out -- what a
    Rover-SLAM Tracking thread sees (SPextractor::operator(), src/Extractors/SPextractor.cc:516-617; SPmatcher::MatchingPoints_onnx(Frame&,
    Frame&), src/Matchers/SPmatcher.cc:457-542; stereo pair of extractor threads, src/Frame.cc:142-147) -- timed by a compiled driver
    (tests/cpp/lat_driver.cpp -> rover-slam_amd/lat_driver, built by __graft_entry__.build()) in a CHILD process, its last results
    checked once against the oracle."""
    import tempfile
    from rover_slam_amd import weights as Wt
    exe = os.path.join(ROOT, "rover-slam_amd", "lat_driver")
    if not os.path.exists(exe):
        return {"error": f"{exe} missing (python -c 'import __graft_entry__ as g; g.build()')"}
    T = 8
    lefts, rights = stereo_frames(synth, T)
    with tempfile.TemporaryDirectory() as d:
        Wt.save(os.path.join(d, "sp.rfew"), wsp, 1)
        Wt.save(os.path.join(d, "lg.rfew"), wlg, 2)
        np.ascontiguousarray(pair_np[:2]).tofile(os.path.join(d, "pair.u8"))
        np.stack([np.stack([l, r]) for l, r in zip(lefts, rights)]).tofile(os.path.join(d, "stereo.u8"))
        env = dict(os.environ, RFE_SP_WEIGHTS=os.path.join(d, "sp.rfew"), RFE_LG_WEIGHTS=os.path.join(d, "lg.rfew"))
        numa = gpu_numa(0)
        if numa and numa.get("local_cpulist") and "RFE_LAT_CPULIST" not in env and os.environ.get("RFE_LAT_NO_PIN") != "1":
            env["RFE_LAT_CPULIST"] = numa["local_cpulist"]       # the driver runs on the CPUs local to the GPU (see lat_driver.cpp)
        if host_graph:
            env["RFE_HOST_GRAPH"] = "1"
        r = subprocess.run([exe, os.path.join(d, "pair.u8"), os.path.join(d, "stereo.u8"), str(T), str(steps), str(warmup), os.path.join(d, "out.bin")],
                           env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            return {"error": f"lat_driver exit {r.returncode}: {(r.stdout + r.stderr)[-300:]}"}
        out = json.loads(r.stdout.strip().splitlines()[-1])
        out["gpu_numa"] = numa
        buf = open(os.path.join(d, "out.bin"), "rb").read()
    out["driver"] = ("tests/cpp/lat_driver.cpp (child process): SPextractor::operator() / SPmatcher::MatchingPoints_onnx(Frame&, Frame&) / left + right "
                     "extractor threads + ComputeStereoMatches_rfe + MatchingPoints_onnx(current, previous); pageable host memory in and out on every call")
    if not check:
        return out
    # ---- the driver's last results against the oracle (once)
    from oracle import oracle as O
    from tolerances import lists_agree_borderline
    off = 0

    def take_frame():
        nonlocal off
        n = int(np.frombuffer(buf, np.int32, 1, off)[0]); off += 4
        kp = np.frombuffer(buf, np.float32, n * 3, off).reshape(n, 3); off += n * 12
        de = np.frombuffer(buf, np.float32, n * 256, off).reshape(n, 256); off += n * 1024
        return n, kp, de

    def take_ints(k):
        nonlocal off
        v = np.frombuffer(buf, np.int32, k, off); off += 4 * k
        return v

    def sp_ok(fr, img):
        n, kp, de = fr
        o = O.superpoint(wsp, img, kmax=KMAX)
        return bool(n == o["n"] and np.array_equal(kp[:, :2], o["kxy"][:n].astype(np.float32)) and np.array_equal(kp[:, 2], o["score"][:n])
                    and np.array_equal(de, o["desc"][:n])), o

    def lg_ok(vn, s, oa, ob, rows, cols):
        lg = O.lightglue(wlg, O.normalize_keypoints(oa["kxy"][:oa["n"]].astype(np.float32), rows, cols),
                         O.normalize_keypoints(ob["kxy"][:ob["n"]].astype(np.float32), rows, cols), oa["desc"][:oa["n"]], ob["desc"][:ob["n"]], debug=True)
        sref = {(int(i), int(j)): float(m) for (i, j), m in zip(lg["pairs"], lg["ms"])}
        got = [(i, int(j)) for i, j in enumerate(vn) if j >= 0]
        ok, _, only = lists_agree_borderline(np.array(got).reshape(-1, 2), [sref.get(k, 0.0) for k in got], lg["pairs"], lg["ms"], lg["scores"], KMAX, tol=tol)
        return bool(ok and s == len(got)), only, lg["S"]

    f0, f1 = take_frame(), take_frame()
    s3, m = take_ints(2)
    vn = take_ints(int(m))
    ok0, o0 = sp_ok(f0, pair_np[0]); ok1, o1 = sp_ok(f1, pair_np[1])
    okm, only3, S3 = lg_ok(vn, int(s3), o0, o1, H, W)
    tcur, tprev = take_ints(2)
    fl, fr_, fp = take_frame(), take_frame(), take_frame()
    (m,) = take_ints(1)
    u_right = np.frombuffer(buf, np.float32, int(m), off); off += 4 * int(m)
    depth = np.frombuffer(buf, np.float32, int(m), off); off += 4 * int(m)
    s5, m = take_ints(2)
    vt = take_ints(int(m))
    okl, ol = sp_ok(fl, lefts[int(tcur)]); okr, orr = sp_ok(fr_, rights[int(tcur)]); okp, op_ = sp_ok(fp, lefts[int(tprev)])
    u_ref, z_ref = O.stereo_match(lefts[int(tcur)], rights[int(tcur)], ol["kxy"][:ol["n"]].astype(np.float32), orr["kxy"][:orr["n"]].astype(np.float32),
                                  ol["desc"][:ol["n"]], orr["desc"][:orr["n"]], 0.11, 0.11 * 435.0)
    oks = bool(np.array_equal(u_right, u_ref) and np.array_equal(depth, z_ref))
    okt, only5, S5 = lg_ok(vt, int(s5), ol, op_, 480, 752)
    out["verified_against_oracle"] = {"ok": bool(ok0 and ok1 and okm and okl and okr and okp and oks and okt),
                                      "c3_superpoint_bit_exact": bool(ok0 and ok1), "c3_match_list_agrees": okm, "c3_matches_oracle": int(S3),
                                      "c3_one_sided_borderline": int(only3), "c5_superpoint_bit_exact": bool(okl and okr and okp),
                                      "c5_stereo_matches_bit_exact": oks, "c5_temporal_match_list_agrees": okt, "c5_matches_oracle": int(S5),
                                      "c5_one_sided_borderline": int(only5)}
    return out


# ------------------------------------------------------------------------------------------------------------------
# multi-rank plumbing
# ------------------------------------------------------------------------------------------------------------------
def spawn_ranks(args, argv):
    """`bench.py --gpus N` without a launcher: start the N ranks as CHILD processes and exit with their status.  Never exec,
    whatever this process did before: a process that has initialised the GPU must not be replaced, and the device census
    below (torch.cuda.device_count() -> amdsmi, or hipGetDeviceCount when amdsmi is absent) may bring the HIP runtime up.
    Rank 0's JSON line goes straight to our stdout."""
    backend = os.environ.get("RFE_BENCH_BACKEND", "nccl")
    if backend == "nccl" and not args.check_launch:
        import torch
        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) are visible; refusing to report a {args.gpus}-GPU number "
                  "(RFE_BENCH_BACKEND=gloo runs the N-rank flow on fewer GPUs as a functional check)", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RFE_BENCH_LAUNCHER="bench.py --gpus (child processes)")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    deadline = None
    while any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() is not None and p.returncode != 0 and deadline is None:
                rc = p.returncode
                deadline = time.time() + 20.0          # a rank died: give the others a moment, then stop them
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()                            # exact PIDs we started
        time.sleep(0.05)
    for p in procs:
        rc = rc or p.returncode
    return rc


def rank_identity(torch, dev, dev_index):
    info = {"rank": int(os.environ.get("RANK", "0")), "pid": os.getpid(), "host": socket.gethostname(), "device_index": dev_index}
    if dev.type == "cuda":
        p = torch.cuda.get_device_properties(dev)
        info.update(name=p.name, arch=getattr(p, "gcnArchName", None), total_memory_gb=round(p.total_memory / 2 ** 30, 1))
        for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"):
            if hasattr(p, k):
                info[k] = int(getattr(p, k))
        if hasattr(p, "uuid"):
            info["uuid"] = str(p.uuid)
    return info


def check_launch(args, rank, world):
    """--check-launch: the N-rank flow without a GPU -- process group, rank census, the ONE-gather result path on CPU
    tensors.  Lets the launcher and the collective plumbing be tested on a box with no (or one) GPU."""
    import torch
    import torch.distributed as dist
    from rover_slam_amd import sharding
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = sharding.shard_frames(4, world, rank) if args.scaling == "weak" else sharding.shard_frames_strong(8 * world, world, rank)
    pack = sharding.ResultPack(shard.frames, 8, torch.device("cpu"))
    pack.n.copy_(torch.arange(shard.start, shard.start + shard.frames, dtype=torch.int32))
    g = sharding.RootGather(pack, world, rank)
    g()
    ident = [None] * world
    me = rank_identity(torch, torch.device("cpu"), -1)
    if world > 1:
        dist.all_gather_object(ident, me)
        ones = torch.ones(1)
        dist.all_reduce(ones)
    else:
        ident, ones = [me], torch.ones(1)
    if rank == 0:
        n_all = sharding.assemble(g, shard.owned)[0]
        print(json.dumps({"check_launch": True, "n_gpus": world, "ranks": world, "scaling": args.scaling,
                          "launcher": os.environ.get("RFE_BENCH_LAUNCHER", "external (torchrun / environment)"),
                          "rccl": {"backend": "gloo", "world_size": world, "allreduce_sum_of_ones": int(ones.item()), "devices": ident},
                          "gathered_frame_ids": n_all.tolist(), "payload_bytes_per_rank": g.nbytes}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    global KMAX, FRAMES_PER_GPU
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): 32 frames per GPU per step; strong: 256 frames per step in all, 256/N per GPU")
    ap.add_argument("--sustained-steps", type=int, default=200,
                    help="further steps after the timed region for the `sustained` object (per-step spread, clocks); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the short Kmax = 256 / 512 and K < Kmax (dustbin weights) runs of SURVEY 8(d) (N=1)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the short PCIe-inclusive measurement (N=1)")
    ap.add_argument("--no-pool", action="store_true", help="skip the short run through the C-ABI pool (rfe_pool_*, N=1)")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the `latency` object: BASELINE configs[1] / [2] / [4] device-resident and through the C++ drop-in classes (N=1)")
    ap.add_argument("--latency-steps", type=int, default=200, help="calls per configuration of the `latency` object (after 20 un-timed ones)")
    ap.add_argument("--gather-desc", action="store_true", help="also gather scores and the 256-d descriptors to rank 0")
    ap.add_argument("--lg-fold", type=int, default=None, choices=[0, 1], help="override RFE_OPT_LG_FOLD_WO (default: the library's)")
    ap.add_argument("--lg-fp16x2", type=int, default=0, choices=[0, 1],
                    help="DIAGNOSTIC runs only (profiling the option): 1 = the whole run with RFE_OPT_LG_FP16X2 on; the line is then labelled as such and is "
                         "not the headline configuration (the default run reports the option as variants.fp16x2)")
    ap.add_argument("--check-launch", action="store_true", help="N-rank flow only (gloo, CPU tensors, no GPU): launcher / collective self-test")
    ap.add_argument("--workload", default="c4", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json configs: c4 (default, the metric's workload) = 33 frames + 32 pairs per GPU; "
                         "c2 = SuperPoint only, batch 1 (latency); c3 = one 640x480 pair, SuperPoint x2 + LightGlue (latency); "
                         "c5 = stereo 752x480 stream: per stereo frame 2 extractions + sparse stereo match + 1 LightGlue match "
                         "of the left image against the previous left image (latency)")
    ap.add_argument("--pairing", default="calibrated", choices=["calibrated", "r04"],
                    help="synthetic weight / frame pairing: calibrated (default; centred descriptor head, calibrated LightGlue law, cell-aligned shifts: hundreds of "
                         "matches per pair, 1e-4 self-check) or r04 (the plain seeded laws and arbitrary shifts of rounds 1-4, 5e-4 self-check)")
    ap.add_argument("--kmax", type=int, default=KMAX, help="keypoint capacity per frame (default 1024)")
    ap.add_argument("--frames-per-gpu", type=int, default=FRAMES_PER_GPU,
                    help="frames (= pairs) each GPU owns per step; 32 = BASELINE configs[3] (256 frames over 8 GPUs), other values are exploratory")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={world}; refusing to report a mislabelled number", file=sys.stderr)
        sys.exit(2)
    if args.check_launch:
        sys.exit(check_launch(args, rank, world))
    if args.workload != "c4" and world > 1:
        print(f"bench.py: --workload {args.workload} is a single-GPU latency configuration; only the metric's workload (c4) shards over ranks",
              file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    backend = os.environ.get("RFE_BENCH_BACKEND", "nccl")   # "gloo": functional check of the N>1 flow on a 1-GPU box
    if backend == "nccl" and world > ndev:
        print(f"bench.py: {world} ranks but {ndev} visible GPU(s)", file=sys.stderr)
        sys.exit(2)
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    rccl = None
    # RFE_BENCH_FORCE_PG=1: build the process group and run every collective of the N-rank flow even at world size 1 -- the
    # RCCL code path (init with device_id, all_gather_object, device all-reduce, the per-step gather) on a 1-GPU box
    pg = world > 1 or os.environ.get("RFE_BENCH_FORCE_PG") == "1"
    if pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints a version banner through C stdio on stdout when the communicator comes up; stdout belongs to the ONE JSON
        # line, so fd 1 points at stderr until the group exists and the C buffers are flushed
        import ctypes
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        # a rank that never arrives must end the run with a message, not hang the driver: every collective of this group times out
        import datetime
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("RFE_BENCH_PG_TIMEOUT_S", "600")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=pg_timeout)
        # what the process group really is: backend, size, one identity record per rank, and a collective on device
        # memory that only comes out right if all `world` ranks took part
        ident = [None] * world
        dist.all_gather_object(ident, rank_identity(torch, dev, dev_index))
        ones = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        rccl = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "allreduce_sum_of_ones": int(ones.item()),
                "devices": ident, "distinct_devices": len({(d.get("pci_bus_id"), d.get("uuid"), d["device_index"]) for d in ident})}
        torch.cuda.synchronize(dev)
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
        if rccl["allreduce_sum_of_ones"] != world:
            print(f"bench.py: all-reduce over the process group saw {rccl['allreduce_sum_of_ones']} ranks, expected {world}", file=sys.stderr)
            sys.exit(3)

    from rover_slam_amd import capi, weights as Wt, synth, sharding
    ctx = capi.Context(dev_index)
    # Weight / frame pairing (round 5).  `calibrated` (default): SuperPoint seed 7 with the centred descriptor head, LightGlue seed 11 with the calibrated
    # law (log-assignment in the range trained weights live in -> the self-check runs at north_star's 1e-4), frames shifted by multiples of the 8-px cell
    # so that an untrained extractor repeats its keypoints: 150-200 matches per pair instead of 3.  `r04`: the pairing of rounds 1-4 (plain seeded laws,
    # arbitrary shifts, ill-conditioned LightGlue logits, 5e-4 bar).  The arithmetic per step is the same (K saturates Kmax either way;
    # `variants.r04_pairing` times the other pairing next to the headline).
    calibrated = args.pairing == "calibrated"
    wsp = Wt.make_superpoint(seed=7, desc_center="auto" if calibrated else None)
    wlg = Wt.make_lightglue(seed=11, calibrated=calibrated)
    from tolerances import LG_SCORE_TOL, LG_SCORE_TOL_CALIBRATED
    lg_tol = LG_SCORE_TOL_CALIBRATED if calibrated else LG_SCORE_TOL
    frame_kw = dict(max_shift=16, shift_step=8) if calibrated else {}
    ctx.set_weights(capi.KIND_SUPERPOINT, wsp)
    ctx.set_weights(capi.KIND_LIGHTGLUE, wlg)
    if args.lg_fold is not None:
        ctx.set_option(capi.OPT_LG_FOLD_WO, args.lg_fold)
    if args.lg_fp16x2:
        ctx.set_option(capi.OPT_LG_FP16X2, 1)
    stream = torch.cuda.Stream(dev)          # library kernels and the RCCL gather share this stream
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    KMAX = args.kmax
    FRAMES_PER_GPU = args.frames_per_gpu
    if args.workload == "c5":
        bench_stereo_stream(args, ctx, capi, synth, torch, dev, rank)
        ctx.close()
        return
    if args.workload != "c4":
        FRAMES_PER_GPU = 1                      # latency configurations: batch 1
    if args.scaling == "strong" and args.workload == "c4":
        shard = sharding.shard_frames_strong(STRONG_TOTAL_FRAMES, world, rank)
        FRAMES_PER_GPU = shard.owned
    else:
        shard = sharding.shard_frames(FRAMES_PER_GPU, world, rank)
    B = shard.frames if args.workload != "c2" else 1
    # ONE common stream of world * owned + 1 frames (configs[3]: 257 for 8 x 32), the same on every rank (seeded, 0.6 s to synthesise);
    # rank r takes frames [owned r, owned r + owned]: its overlap frame IS rank r + 1's first frame, no inter-GPU dependency.
    # At N = 1 this is the 33-frame stream of every earlier round.
    stream_np, _ = synth.make_frames(world * shard.owned + 1, H, W, seed=20240314, **frame_kw)
    frames_np = np.ascontiguousarray(stream_np[shard.start:shard.start + B])
    del stream_np
    frames = torch.from_numpy(frames_np).to(dev)
    pack = sharding.ResultPack(B, KMAX, dev)    # every output of the step in one contiguous buffer; the gather moves its prefix
    n, kxy, score, desc, S, pairs, ms = pack.n, pack.kxy, pack.score, pack.desc, pack.S, pack.pairs, pack.ms
    gather = sharding.RootGather(pack, world, rank, with_desc=args.gather_desc, always_collective=pg) if pg else None

    def step():
        if args.workload == "c2":
            ctx._chk(capi.lib.rfe_extract_u8_dev(ctx.h, frames.data_ptr(), H, W, W, 1, KMAX, 0.0005, n.data_ptr(), kxy.data_ptr(),
                                                 score.data_ptr(), desc.data_ptr()))
            return
        ctx._chk(capi.lib.rfe_extract_match_stream_dev(
            ctx.h, frames.data_ptr(), H, W, W, B, KMAX, 0.0005, 0.1, n.data_ptr(), kxy.data_ptr(), score.data_ptr(),
            desc.data_ptr(), S.data_ptr(), pairs.data_ptr(), ms.data_ptr()))
        if gather is not None:   # the trivial gather over RCCL/xGMI: ONE collective, preallocated receive buffer on rank 0
            gather()

    def fence():
        if pg:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    collective_warmup = 0
    if pg:
        # the first collectives of a communicator pay connection set-up (RCCL: channel / proxy bring-up per peer); whatever --warmup says,
        # the timed region never holds them: three more un-timed rounds of the step's own gather + the fence's barrier
        if args.warmup == 0:
            step()
        for _ in range(3):
            gather()
            fence()
            collective_warmup += 1
    # Stage table: a separate, untimed pass with events around EVERY stage (each event pair keeps neighbouring kernels
    # from overlapping their launch / drain: ~2 % of a batched step).  It also names the dominant kernel.
    full_steps = args.steps if args.workload != "c4" else min(args.steps, 3)
    ctx.profile_filter(None)
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(full_steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof_full = ctx.profile_read()
    ctx.profile(False)
    prof_dom = None
    if args.workload == "c4":
        # Timed region: exactly K steps, HIP events only around the dominant kernel (its average launch duration feeds
        # the roofline object and is measured here, live, on the stream the kernels run on).
        dom_stage = max(prof_full.items(), key=lambda kv: kv[1][0])[0]
        ctx.profile_filter(dom_stage)
        ctx.profile(True)
        ctx.profile_reset()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        prof_dom = ctx.profile_read()
        ctx.profile(False)
        ctx.profile_filter(None)
    prof = prof_full
    per_rank_ms = [dt / max(args.steps, 1) * 1e3]
    if pg:
        tall = torch.zeros(world, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        tall[rank] = dt
        dist.all_reduce(tall)                      # every rank's own clock around the same barrier-bracketed region
        per_rank_ms = [float(v) / args.steps * 1e3 for v in tall.tolist()]
        dt = float(tall.max().item())              # MAX over ranks

    if args.workload != "c4":      # latency configurations: a short line, no roofline object (not the metric's workload)
        # the per-stage HIP events cost a few microseconds each, which shows at these step times: the headline of a
        # latency run is a second timed loop without them, the stage table comes from the profiled loop above
        dt_prof = dt
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        if rank == 0:
            unit = "frames/s" if args.workload == "c2" else "pairs/s"
            print(json.dumps({"metric": f"BASELINE configs[{1 if args.workload == 'c2' else 2}] latency run", "value": round(args.steps / dt, 2),
                              "unit": unit, "ms_per_step": round(dt / args.steps * 1e3, 4), "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "kmax": KMAX, "mean_keypoints": float(n.float().mean().item()),
                              "ms_per_step_with_stage_events": round(dt_prof / args.steps * 1e3, 4),
                              "stages_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}))
        ctx.close()
        return

    # ---- sustained: SURVEY 8(d) asks for >= 200 timed iterations.  The headline keeps the driver's --steps; this loop
    # runs the same step (gather included) `--sustained-steps` more times with one event per step on the stream (no host
    # synchronisation inside the loop) and reports the spread and the clocks the chip settled at.
    sustained = None
    if args.sustained_steps > 0:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.sustained_steps + 1)]
        fence()
        clk0 = gpu_clocks(dev_index)
        t0 = time.perf_counter()
        evs[0].record(stream)
        for i in range(args.sustained_steps):
            step()
            evs[i + 1].record(stream)
        clk_mid = gpu_clocks(dev_index)             # read while the queue is still draining
        fence()
        wall = time.perf_counter() - t0
        per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(args.sustained_steps)], np.float64)
        wmax = wall
        if pg:
            tw = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            wmax = float(tw.item())
        sustained = {"steps": args.sustained_steps, "seconds": round(wmax, 3),
                     "value": round(FRAMES_PER_GPU * world * args.sustained_steps / wmax, 2), "unit": "frames/s",
                     "ms_per_step": {"mean": round(float(per.mean()), 4), "std": round(float(per.std()), 4), "min": round(float(per.min()), 4),
                                     "max": round(float(per.max()), 4), "p50": round(float(np.median(per)), 4),
                                     "first_10_mean": round(float(per[:10].mean()), 4), "last_10_mean": round(float(per[-10:].mean()), 4)},
                     "clocks_before": clk0, "clocks_under_load": clk_mid,
                     "note": "rank 0's per-step HIP-event intervals on the library's stream; value = all ranks' frames / max-over-ranks wall time"}

    # ---- variants ("also report K in {256, 512}" and "a second weight set with dustbin bias to exercise K < Kmax and
    # variable-K batching"): the same 33-frame / 32-pair step at other keypoint budgets, short runs, never the headline
    variants = None
    fp16x2_out = None
    perf_notes = {}        # timing observations of auxiliary runs: information only, never an exit status
    aux_mismatch = []      # CORRECTNESS mismatches of auxiliary runs (never swallowed like infrastructure failures): -> out["invalid_aux"], exit status 5
    if world == 1 and not args.no_variants and args.workload == "c4" and not args.lg_fp16x2:
        variants = {}
        try:

            def run_variant(kmax, tag, note, nst=None):
                vp = sharding.ResultPack(B, kmax, dev)
                def vstep():
                    ctx._chk(capi.lib.rfe_extract_match_stream_dev(
                        ctx.h, frames.data_ptr(), H, W, W, B, kmax, 0.0005, 0.1, vp.n.data_ptr(), vp.kxy.data_ptr(), vp.score.data_ptr(),
                        vp.desc.data_ptr(), vp.S.data_ptr(), vp.pairs.data_ptr(), vp.ms.data_ptr()))
                for _ in range(2):
                    vstep()
                fence()
                nst = nst or max(3, min(args.steps, 10))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t1 = time.perf_counter()
                e0.record(stream)                  # the library's stream: device time of the loop next to the host's wall clock
                for _ in range(nst):
                    vstep()
                e1.record(stream)
                fence()
                dtv = time.perf_counter() - t1
                kk = vp.n.cpu().numpy()
                variants[tag] = {"value": round(FRAMES_PER_GPU * nst / dtv, 2), "unit": "frames/s", "kmax": kmax, "steps": nst,
                                 "ms_per_step": round(dtv / nst * 1e3, 3), "ms_per_step_hip_events": round(e0.elapsed_time(e1) / nst, 3),
                                 "keypoints_per_frame": {"mean": float(kk.mean()), "min": int(kk.min()), "max": int(kk.max())},
                                 "matches_per_pair_mean": float(vp.S.float().mean().item()), "note": note}
            # the pairing of rounds 1-4 next to the headline: same arithmetic, other weights / frames -> the step time must not move
            if calibrated:
                # A PERFORMANCE NOTE, never a correctness verdict: both sides are the same 10-step loop (no collective on either), timed with HIP events on the
                # library's stream; reported under out["perf_notes"], and only when the process runs alone on its stream's clock (no process group)
                run_variant(KMAX, "headline_same_loop", "the headline pairing through the variant loop (10 steps, no gather): the like-for-like partner of r04_pairing", nst=10)
                fr_keep = frames.clone()
                try:
                    ctx.set_weights(capi.KIND_SUPERPOINT, Wt.make_superpoint(seed=7)); ctx.set_weights(capi.KIND_LIGHTGLUE, Wt.make_lightglue(seed=11))
                    old_np, _ = synth.make_frames(world * shard.owned + 1, H, W, seed=20240314)
                    frames.copy_(torch.from_numpy(np.ascontiguousarray(old_np[shard.start:shard.start + B])).to(dev))
                    run_variant(KMAX, "r04_pairing", "the weight / frame pairing of rounds 1-4 (plain seeded laws, shifts of -8..8 px): same step, 2-4 matches per pair", nst=10)
                finally:     # whatever happened: the bench state (frames, weights) is the headline's again before anything else reads it
                    frames.copy_(fr_keep); del fr_keep
                    ctx.set_weights(capi.KIND_SUPERPOINT, wsp); ctx.set_weights(capi.KIND_LIGHTGLUE, wlg)
                if not pg:
                    a_, b_ = variants["r04_pairing"]["ms_per_step_hip_events"], variants["headline_same_loop"]["ms_per_step_hip_events"]
                    perf_notes["r04_pairing_step_time_ratio"] = {"value": round(a_ / b_, 4), "r04_ms": a_, "headline_ms": b_, "steps_each": 10,
                                                                 "clock": "HIP events on the library's stream, same loop on both sides",
                                                                 "note": "informational: same arithmetic, other weights / frames; expected within a few percent"}
            # configs[3] under STRONG scaling at N = 1: all 257 frames of the node's batch in ONE call (what `--scaling strong --gpus 1` times), next to the
            # weak-scaling headline: the 1-GPU number must not depend on the 33-frame batch
            try:
                nF = STRONG_TOTAL_FRAMES + 1
                big_np, _ = synth.make_frames(nF, H, W, seed=20240314, **frame_kw)
                big = torch.from_numpy(big_np).to(dev)
                bp = sharding.ResultPack(nF, KMAX, dev)
                def bstep():
                    ctx._chk(capi.lib.rfe_extract_match_stream_dev(ctx.h, big.data_ptr(), H, W, W, nF, KMAX, 0.0005, 0.1, bp.n.data_ptr(), bp.kxy.data_ptr(),
                                                                   bp.score.data_ptr(), bp.desc.data_ptr(), bp.S.data_ptr(), bp.pairs.data_ptr(), bp.ms.data_ptr()))
                bstep(); fence()
                t1 = time.perf_counter()
                for _ in range(3):
                    bstep()
                fence()
                dtb = (time.perf_counter() - t1) / 3
                # the first 33 frames ARE the headline batch: same keypoints, descriptors and match lists bit for bit
                same = bool(torch.equal(bp.n[:B], n) and torch.equal(bp.kxy[:B], kxy) and torch.equal(bp.desc[:B], desc) and torch.equal(bp.S[:B - 1], S[:B - 1])
                            and all(torch.equal(bp.pairs[q, :int(S[q])], pairs[q, :int(S[q])]) and torch.equal(bp.ms[q, :int(S[q])], ms[q, :int(S[q])]) for q in range(B - 1)))
                variants["strong_n1"] = {"value": round(STRONG_TOTAL_FRAMES / dtb, 2), "unit": "frames/s", "frames_per_call": nF, "pairs_per_call": nF - 1, "ms_per_step": round(dtb * 1e3, 3),
                                         "first_33_frames_equal_headline_batch": same, "workspace_bytes": int(ctx.workspace_bytes()),
                                         "note": "configs[3] with --scaling strong at N = 1: 257 frames extracted + 256 pairs matched by ONE rfe_extract_match_stream_dev call"}
                if not same:
                    aux_mismatch.append("variants.strong_n1: the first 33 frames of the 257-frame call differ from the headline batch")
                del big, bp
            except capi.RfeError as e:
                variants["strong_n1"] = {"error": str(e)[:300]}
            run_variant(512, "kmax512", "same frames and weights, keypoint budget 512")
            run_variant(256, "kmax256", "same frames and weights, keypoint budget 256")
            # bias chosen on the bench frames with the oracle: +9 still saturates Kmax = 1024 on every frame, +9.5 leaves 700-900 keypoints
            # (a different count per frame), +10 about 340, +12 none
            ctx.set_weights(capi.KIND_SUPERPOINT, Wt.make_superpoint(seed=7, dustbin_bias=9.5))
            run_variant(KMAX, "dustbin_k_below_kmax", "SuperPoint weights with dustbin bias +9.5 (SURVEY 8(d)): fewer candidates than Kmax pass the 0.0005 "
                                                      "threshold, every frame has its own keypoint count (ragged sequences, masked attention / assignment)")
            ctx.set_weights(capi.KIND_SUPERPOINT, wsp)
            step(); fence()      # the resident results are those of the bench weights again (cpu_baseline checks them)
            # RFE_OPT_LG_FP16X2 (default off, include/rover_fe.h): LightGlue's Linears and attention as split products on the f16 matrix pipe (gemm_h2.hip, lg_attention_h2.hip).
            # Same frames, weights and Kmax as the headline; compared here with the fp32 path's resident results, never the headline.
            S32, p32, m32 = S.cpu().numpy().copy(), pairs.cpu().numpy().copy(), ms.cpu().numpy().copy()
            ctx.set_option(capi.OPT_LG_FP16X2, 1)
            try:
                run_variant(KMAX, "fp16x2", "RFE_OPT_LG_FP16X2 = 1: every LightGlue Linear and the fused attention of the batched call as fp16 (hi + lo) x fp16 (hi + lo), "
                                            "three products on v_mfma_f32_32x32x16_f16 with fp32 accumulation; softmax, LayerNorm / GELU, assignment and SuperPoint unchanged (fp32)")
                ctx.profile_filter(None); ctx.profile(True); ctx.profile_reset()
                for _ in range(3):
                    step()
                fence()
                ph2 = ctx.profile_read()
                ctx.profile(False)
                variants["fp16x2"]["stages_ms_per_step"] = {k: round(v[0] / 3, 4) for k, v in sorted(ph2.items(), key=lambda kv: -kv[1][0])
                                                                    if k.startswith("lg_")}
                Sh, ph, mh = S.cpu().numpy(), pairs.cpu().numpy(), ms.cpu().numpy()
                same = bool(np.array_equal(Sh, S32) and all(np.array_equal(ph[q, :S32[q]], p32[q, :S32[q]]) for q in range(B - 1)))
                devh = max([float(np.abs(mh[q, :S32[q]] - m32[q, :S32[q]]).max()) for q in range(B - 1) if S32[q] > 0 and Sh[q] == S32[q]] or [0.0]) if same else None
                if not same:     # informational for the option (its own oracle check follows in cpu_baseline), but never silent
                    print("bench.py: variants.fp16x2 match lists differ from the fp32 path's", file=sys.stderr)
                variants["fp16x2"].update({"match_lists_identical_to_fp32_path": same, "match_score_max_dev_vs_fp32_path": devh,
                                                   "matches_total": int(S32.sum())})
                fp16x2_out = {"S": Sh.copy(), "pairs": ph.copy(), "ms": mh.copy()}
            finally:
                ctx.set_option(capi.OPT_LG_FP16X2, 0)
            step(); fence()      # ... and of the fp32 path
        except Exception as e:     # an auxiliary run must never take the headline down with it: record the failure, restore the bench state
            variants["error"] = f"{type(e).__name__}: {e}"[:300]
            print(f"bench.py: variants failed: {variants['error']}", file=sys.stderr)
            ctx.set_option(capi.OPT_LG_FP16X2, 0)
            ctx.set_weights(capi.KIND_SUPERPOINT, wsp)
            fp16x2_out = None
            step(); fence()

    pcie = None
    if world == 1 and not args.no_pcie:
        try:
            # PCIe-inclusive variant (never the headline `value`): frames start in pinned host memory, results end there.
            # Double buffered: a copy stream uploads batch k+1 and downloads the results of batch k-1 while batch k computes.
            h_frames = torch.from_numpy(frames_np).pin_memory()
            cstream = torch.cuda.Stream(dev)
            sets = []
            for _ in range(2):
                dv = [torch.zeros_like(t) for t in (frames, n, kxy, score, desc, S, pairs, ms)]   # match lists are written up to S only
                hv = [torch.empty_like(t, device="cpu").pin_memory() for t in dv[1:]]
                sets.append((dv, hv, torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()))
            def step_pcie(k):
                dv, hv, ev_up, ev_done, ev_down = sets[k % 2]
                with torch.cuda.stream(cstream):
                    cstream.wait_event(ev_done)                    # batch k-2 no longer reads this frame buffer ...
                    dv[0].copy_(h_frames, non_blocking=True)
                    ev_up.record(cstream)
                stream.wait_event(ev_up)
                stream.wait_event(ev_down)                         # ... and its results have left the device
                f_, n_, k_, s_, d_, S_, p_, m_ = dv
                ctx._chk(capi.lib.rfe_extract_match_stream_dev(
                    ctx.h, f_.data_ptr(), H, W, W, B, KMAX, 0.0005, 0.1, n_.data_ptr(), k_.data_ptr(), s_.data_ptr(),
                    d_.data_ptr(), S_.data_ptr(), p_.data_ptr(), m_.data_ptr()))
                ev_done.record(stream)
                with torch.cuda.stream(cstream):
                    cstream.wait_event(ev_done)
                    for h, t in zip(hv, dv[1:]):
                        h.copy_(t, non_blocking=True)
                    ev_down.record(cstream)
            for k in range(2):
                step_pcie(k)
            fence()
            t1 = time.perf_counter()
            npc = max(4, min(args.steps, 10))
            for k in range(npc):
                step_pcie(k)
            fence()
            pcie = FRAMES_PER_GPU * npc / (time.perf_counter() - t1)
            for si, (dv_, hv_, *_e) in enumerate(sets):               # same results as the resident path
                for nm, h_, t_ in zip(("n", "kxy", "score", "desc", "S", "pairs", "ms"), hv_, (n, kxy, score, desc, S, pairs, ms)):
                    if not torch.equal(h_, t_.cpu()):
                        raise AuxMismatch(f"PCIe pipeline: {nm} of buffer set {si} differs from the resident path "
                                          f"({int((h_ != t_.cpu()).sum())} elements)")
        except AuxMismatch as e:
            pcie = {"error": str(e)[:300], "mismatch": True}
            aux_mismatch.append(f"pcie_inclusive: {e}"[:300])
            print(f"bench.py: {e}", file=sys.stderr)
            fence()
        except Exception as e:
            pcie = {"error": f"{type(e).__name__}: {e}"[:300]}
            print(f"bench.py: PCIe-inclusive run failed: {pcie['error']}", file=sys.stderr)
            fence()

    pool_line = None
    if world == 1 and not args.no_pool and args.workload == "c4":
        # The same 33-frame step through the C-ABI pool a C / C++ host would use (rfe_pool_*, rover-slam_amd/csrc/rfe_pool.hip): HOST frames in,
        # HOST results out (pageable memory, no overlap between calls), one member on this device, its rows gathered into the root buffer
        # through RCCL when librccl could be opened (self send / receive), copies otherwise.  Never the headline value.
        # (RCCL's version banner goes to C stdout when the pool's communicator comes up: fd 1 points at stderr for the duration)
        import ctypes as _ctp
        sys.stdout.flush()
        _saved1 = os.dup(1)
        os.dup2(2, 1)
        pool = None
        try:
            pool = capi.Pool([dev.index or 0])
            pool.set_weights(capi.KIND_SUPERPOINT, wsp); pool.set_weights(capi.KIND_LIGHTGLUE, wlg)
            if args.lg_fold is not None:
                pool.set_option(capi.OPT_LG_FOLD_WO, args.lg_fold)
            tr = capi.POOL_RCCL if pool.has_rccl else capi.POOL_COPY
            po = pool.extract_match_stream(frames_np, kmax=KMAX, transport=tr, with_desc=False)
            t1 = time.perf_counter()
            npl = 5
            for _ in range(npl):
                po = pool.extract_match_stream(frames_np, kmax=KMAX, transport=tr, with_desc=False)
            dtp = time.perf_counter() - t1
            S_h, p_h, m_h = S.cpu().numpy(), pairs.cpu().numpy(), ms.cpu().numpy()
            same = bool(np.array_equal(po["n"], n.cpu().numpy()) and np.array_equal(po["kxy"], kxy.cpu().numpy()) and np.array_equal(po["S"], S_h)
                        and all(np.array_equal(po["pairs"][q, :S_h[q]], p_h[q, :S_h[q]]) and np.array_equal(po["ms"][q, :S_h[q]], m_h[q, :S_h[q]])
                                for q in range(B - 1)))
            pool_line = {"value": round(FRAMES_PER_GPU * npl / dtp, 2), "unit": "frames/s", "members": pool.size,
                         "transport": "rccl (grouped ncclSend / ncclRecv into the root buffer)" if tr == capi.POOL_RCCL else "copy",
                         "equals_resident_path": same,
                         "note": "rfe_pool_extract_match_stream: frames from pageable host memory, counts / keypoints / matches back to host arrays, "
                                 "calls not overlapped; the C-ABI route to configs[3], not the headline value"}
            if not same:
                aux_mismatch.append("pool_c_abi: the pool call's results differ from the resident path")
                print("bench.py: the pool call's results differ from the resident path", file=sys.stderr)
        except Exception as e:
            pool_line = {"error": f"{type(e).__name__}: {e}"[:300]}
            print(f"bench.py: pool run failed: {pool_line['error']}", file=sys.stderr)
        finally:
            if pool is not None:
                pool.close()
            _ctp.CDLL(None).fflush(None)
            os.dup2(_saved1, 1)
            os.close(_saved1)

    latency = None
    if world == 1 and not args.no_latency and args.workload == "c4" and not args.lg_fp16x2:
        # The reference's own call pattern (batch 1) on the driver's clock: configs[1] / [2] / [4], device-resident and through the C++
        # drop-in classes with host pointers.  Never the headline value.
        latency = {"note": "BASELINE configs[1] / [2] / [4]: batch-1 latency in ms per call (c2 = one frame, c3 = one pair incl. both extractions, "
                           "c5 = one 752x480 stereo frame); `resident` = device-resident entry points, `dropin` = the C++ drop-in classes with "
                           "host pointers, as a Rover-SLAM thread calls them"}
        try:
            latency["resident"] = latency_resident(ctx, capi, synth, sharding, torch, dev, frames, steps=args.latency_steps,
                                                   check=None if args.no_cpu_baseline else (frames_np, wsp, wlg, lg_tol))
        except Exception as e:
            latency["resident"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        try:
            # >= 500 calls each, per-call samples: a Tracking thread sees single calls (p50 / p95 / max), not a mean
            latency["dropin"] = latency_dropin(synth, wsp, wlg, frames_np, steps=max(args.latency_steps, 500), warmup=20, check=not args.no_cpu_baseline,
                                               tol=lg_tol)
        except Exception as e:
            latency["dropin"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        try:    # the same driver with RFE_OPT_HOST_GRAPH on (default off): every host entry submits ONE replayed hipGraph per call; results checked the same way
            hg = latency_dropin(synth, wsp, wlg, frames_np, steps=max(args.latency_steps, 500), warmup=20, check=not args.no_cpu_baseline, tol=lg_tol, host_graph=True)
            latency["dropin_host_graph"] = {k: hg[k] for k in ("c2_ms", "c3_ms", "c5_ms", "per_call_ms", "verified_against_oracle", "error") if k in hg}
            latency["dropin_host_graph"]["note"] = "RFE_OPT_HOST_GRAPH = 1 (include/rover_fe.h): a tail-latency option, not the default; the device is busy 98 % of a one-pair call either way"
        except Exception as e:
            latency["dropin_host_graph"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        # the same device-resident calls with RFE_OPT_LG_FP16X2 on (default off, never the headline): at one pair the Linears take the split form of
        # the latency tiles (gemm_lat.hip, H2) and the attention the split form of lg_attention_lat.hip.  SuperPoint (c2) is unaffected and not repeated.
        try:
            ctx.set_option(capi.OPT_LG_FP16X2, 1)
            r2 = latency_resident(ctx, capi, synth, sharding, torch, dev, frames, steps=args.latency_steps,
                                  check=None if args.no_cpu_baseline else (frames_np, wsp, wlg, lg_tol))
            latency["resident_fp16x2"] = {"c3": r2["c3"], "c5": r2["c5"], **({"verified_against_oracle": r2["verified_against_oracle"]} if "verified_against_oracle" in r2 else {}),
                                          "note": "RFE_OPT_LG_FP16X2 = 1: LightGlue's Linears AND the one-pair attention as fp16 hi + lo split products (three f16 matrix "
                                                  "instructions per fp32 product, fp32 accumulation); match lists / scores of this configuration are oracle-checked by "
                                                  "tests/test_gpu_throughput_parity.py::test_lightglue_one_pair_fp16x2_vs_oracle"}
        except Exception as e:
            latency["resident_fp16x2"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            ctx.set_option(capi.OPT_LG_FP16X2, 0)
        for k in ("resident", "dropin", "dropin_host_graph", "resident_fp16x2"):
            if "error" in latency[k]:
                print(f"bench.py: latency.{k} failed: {latency[k]['error']}", file=sys.stderr)
        step(); fence()      # the resident results are those of the bench batch again (cpu_baseline checks them)

    bad_exit = 0
    gathered_ok = None
    if pg and rank == 0:   # the gathered payload of the last step really holds every rank's results
        gn = sharding.assemble(gather, shard.owned)[0]
        gathered_ok = bool(gn.numel() == world * shard.owned + 1 and torch.equal(gather.rank_view(0, "n").cpu(), n.cpu()) and int((gn > 0).sum()) == gn.numel())

    if rank == 0:
        lens = n.cpu().numpy()
        total_frames = FRAMES_PER_GPU * world * args.steps
        value = total_frames / dt
        # dominant kernel = the stage with the largest accumulated time in the full pass; its events come from the timed region
        dom_name, (dom_ms, dom_calls) = next(iter(prof_dom.items()))
        fl = stage_flops(dom_name, B, lens, dom_calls // args.steps)
        avg_ms = dom_ms / max(dom_calls, 1)
        achieved = fl / (avg_ms * 1e-3) / 1e12 if fl else None
        traffic, traffic_src = pmc_traffic(dom_name)
        stages = {}
        for k, (msv, calls) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
            f = stage_flops(k, B, lens, calls // full_steps)
            stages[k] = {"ms_per_step": round(msv / full_steps, 4), "launches_per_step": calls // full_steps,
                         "tflops": round(f / (msv / calls * 1e-3) / 1e12, 2) if f else None}
        fold = ctx.get_option(capi.OPT_LG_FOLD_WO)
        out = {
            "metric": "frames/s SuperPoint+LightGlue 640x480", "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[3] per GPU: {FRAMES_PER_GPU + 1} synthetic 640x480 u8 frames resident in HBM, SuperPoint extract "
                                   f"(Kmax={KMAX}, thr=0.0005) + LightGlue match of {FRAMES_PER_GPU} consecutive pairs (9 layers, filter 0.1); "
                                   f"{FRAMES_PER_GPU} frames counted per GPU per step; seeded synthetic weights ({'centred descriptor head + calibrated LightGlue law, frames shifted by multiples of 8 px' if calibrated else 'plain laws of rounds 1-4, arbitrary shifts'}); the same resident frames every step "
                                   "(compute-bound path, no data-dependent control flow besides the keypoint counts)",
                       "frames_per_gpu": FRAMES_PER_GPU, "kmax": KMAX, "mean_keypoints": float(lens.mean()), "lg_fold_wo": fold, "pairing": args.pairing,
                       "matches_per_pair_mean": float(S.float().mean().item()),
                       "sharding": f"frames sharded over {world} GPU(s), 1 overlap frame per rank"
                                   + ("; ONE gather per step of counts/keypoints/matches to rank 0 over RCCL" if world > 1 else "")},
            "ranks": world,
            "launcher": os.environ.get("RFE_BENCH_LAUNCHER", "external (torchrun / environment)" if world > 1 else "single process"),
            "per_rank_ms_per_step": {"min": round(min(per_rank_ms), 3), "max": round(max(per_rank_ms), 3), "all": [round(v, 3) for v in per_rank_ms],
                                     "outlier": bool(max(per_rank_ms) > 1.05 * min(per_rank_ms)),
                                     "outlier_rule": "max / min > 1.05: one rank's own clock over the barrier-bracketed region is 5 % off the fastest (a slow GPU, a noisy host core)"},
            "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 2) if achieved else None,
                         "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_TFLOPS, 4) if achieved else None,
                         "traffic": traffic, "traffic_unit": "bytes/launch (rocprofv3 PMC, separate pass)",
                         "traffic_source": traffic_src, "avg_launch_ms": round(avg_ms, 4), "launches": dom_calls,
                         "flops_per_launch": fl},
            "stages": stages,
            "stages_note": f"separate untimed pass of {full_steps} steps with events around every stage (costs ~2 %); the timed "
                           "region instruments the dominant kernel only",
        }
        if args.lg_fp16x2:
            out["dtype"] = "f32; LightGlue Linears + attention as fp16 hi + lo split products on the f16 matrix pipe, fp32 accumulation (RFE_OPT_LG_FP16X2)"
            out["diagnostic"] = "--lg-fp16x2 1: a profiling run of the OPTION, not the headline configuration (default runs keep the option off and report it as variants.fp16x2)"
            if dom_name.startswith("lg_") and achieved:
                pk = 2500.0 / 3.0   # f16 dense peak / three products per fp32 product
                out["roofline"].update({"peak": round(pk, 1), "frac": round(achieved / pk, 4), "unit": "TFLOP/s fp32-equivalent"})
        if rccl is not None:
            out["rccl"] = rccl
            out["rccl"]["untimed_collective_warmup_rounds"] = collective_warmup
            out["gather"] = {"collectives_per_step": 1, "payload_bytes_per_rank": gather.nbytes, "with_descriptors": bool(args.gather_desc),
                             "receive_buffer": "preallocated once on rank 0", "last_step_payload_verified": gathered_ok}
        if sustained is not None:
            out["sustained"] = sustained
        if variants is not None:
            out["variants"] = variants
        if pool_line is not None:
            out["pool_c_abi"] = pool_line
        if latency is not None:
            out["latency"] = latency
        if isinstance(pcie, dict):
            out["pcie_inclusive"] = pcie
        elif pcie is not None:
            out["pcie_inclusive"] = {"value": round(pcie, 2), "unit": "frames/s",
                                     "note": "same step with H2D of the 33 u8 frames and D2H of all results (descriptors included) per step, pinned host memory, double buffered on a copy stream; not the headline value"}
        if world == 1 and not args.no_cpu_baseline:
            gpu = {k: v.cpu().numpy() for k, v in (("n", n), ("kxy", kxy), ("score", score), ("desc", desc), ("S", S), ("pairs", pairs), ("ms", ms))}
            gpu["fold"] = fold
            if fp16x2_out is not None:
                gpu["alt"] = fp16x2_out
            out["cpu_baseline"] = cpu_baseline(frames_np, wsp, wlg, gpu, tol=lg_tol)
            vf = (out["cpu_baseline"].get("verified_against_gpu") or {}).pop("fp16x2_variant", None)
            if vf is not None and variants is not None and "fp16x2" in variants:
                variants["fp16x2"]["verified_against_oracle"] = vf
            out["cpu_baseline_torch"] = torch_cpu_baseline(frames_np, wsp, wlg)
            ref = ort_reference_baseline(frames_np)          # only where onnxruntime + the real blobs exist (not in this image)
            if ref is not None:
                out["cpu_baseline_port"] = out["cpu_baseline"]
                out["cpu_baseline"] = ref
        try:        # anything a native library still holds in C stdio goes out BEFORE the JSON line
            import ctypes as _ct
            _ct.CDLL(None).fflush(None)
        except OSError:
            pass
        if perf_notes:
            out["perf_notes"] = perf_notes
        bad_exit = exit_status(out, aux_mismatch)
        if bad_exit:
            print(f"bench.py: exit {bad_exit}: {out.get('invalid') or ''} {out.get('invalid_aux') or ''}", file=sys.stderr)
        print(json.dumps(out), flush=True)
    ctx.close()
    if pg:
        dist.barrier()
        dist.destroy_process_group()
    if bad_exit:
        sys.exit(bad_exit)


if __name__ == "__main__":
    main()
