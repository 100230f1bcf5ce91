#!/usr/bin/env python3
"""Graph-execution fixtures: tests/golden/onnx_s0.npz, onnx_s1.npz.

For each of two settings the published SuperPoint WITH ITS REAL TAIL and the fused LightGlue are exported by torch's ONNX serialiser
(tools/onnx_export.py) with this repo's seeded weights and the reference's tensor names, the graph FILES are executed by
tools/mini_onnx.py through tools/ort_parity.py (which also requires the oracle to agree: exit code 0), and inputs plus every graph
output are written as a fixture.  `-m gpu` tests replay them against librover_fe.so (tests/test_gpu_onnx_graph.py: weights and
hyper-parameters go .onnx -> RFEW v2 -> rfe_load_weights); tests/test_ort_parity.py checks on CPU that a fresh execution still gives
the committed fixture.  Build-container tool; nothing of the reference is read (its .onnx files are missing, .MISSING_LARGE_BLOBS:4-5).

  s0: NMS radius 4, border 4, threshold 0.0005, K = 1024 (top-k cut active), 240 x 320, filter 0.1, CALIBRATED LightGlue weights -> 1e-4 bar
  s1: NMS radius 3, border 2, threshold 0.005, K = 2048 (every candidate kept, ordered), 120 x 160, filter 0.25, seeded (ill-conditioned)
      LightGlue weights -> tests/tolerances.py LG_SCORE_TOL

    python tools/gen_onnx_golden.py
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "s0": dict(setting=0, lg_thr=0.1, calibrated=True, size=(240, 320), frames=2, mscore_tol=1e-4, min_matches=100),
    "s1": dict(setting=1, lg_thr=0.25, calibrated=False, size=(120, 160), frames=2, mscore_tol=5e-4, min_matches=30),
}


def export_case(d, case):
    """-> (superpoint.onnx, lightglue_sim.onnx) written under d for CASES[case]"""
    import onnx_export as X
    c = CASES[case]
    sp, _ = X.export_sp(d, X.SETTINGS[c["setting"]], seed=7, desc_center="auto")
    lg, _ = X.export_lg(d, c["lg_thr"], seed=11, calibrated=c["calibrated"])
    return sp, lg


def harness_args(case, sp, lg):
    c = CASES[case]
    return ["--superpoint", sp, "--lightglue", lg, "--frames", str(c["frames"]), "--height", str(c["size"][0]), "--width", str(c["size"][1]),
            "--shift-step", "8", "--mscore-tol", str(c["mscore_tol"]), "--min-matches", str(c["min_matches"])]


def main():
    import ort_parity
    for case in CASES:
        with tempfile.TemporaryDirectory() as d:
            sp, lg = export_case(d, case)
            out = os.path.join(ROOT, "tests", "golden", f"onnx_{case}.npz")
            rc = ort_parity.main(harness_args(case, sp, lg) + ["--backend", "mini", "--save", out])
            if rc != 0:
                sys.exit(f"gen_onnx_golden: {case}: the oracle does not agree with the graph execution (exit {rc}); no fixture kept")
            print(f"{out}: {os.path.getsize(out)} bytes")


if __name__ == "__main__":
    main()
