"""-m gpu: librover_fe.so against EXECUTIONS OF THE EXPORTED GRAPH FILES.

tests/golden/onnx_s0.npz / onnx_s1.npz hold what tools/mini_onnx.py computed from superpoint.onnx / lightglue_sim.onnx files written by
torch's ONNX exporter (tools/gen_onnx_golden.py; the stand-in this image allows for `Session::Run`,
src/Extractors/superpoint_onnx.cc:133-136, src/Matchers/lightglue_onnx.cpp:210-214).  The same graph files are re-exported on the GPU
box (same seeds; the fixture's weight hash must match), converted by onnx_weights into RFEW v2 containers -- weights AND the graph's
hyper-parameters -- loaded with rfe_load_weights, and tools/ort_parity.py --gpu replays the recorded graph outputs against the HIP
path: identical keypoint sets / order, scores, descriptors <= 1e-4, identical match lists, match scores <= 1e-4 on the calibrated
LightGlue weights (s0) / the stated 5e-4 on the ill-conditioned seeded set (s1)."""
import os

import pytest

torch = pytest.importorskip("torch")

import gen_onnx_golden as G  # noqa: E402
import onnx_export as X  # noqa: E402
import ort_parity as P  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["s0", "s1"])
def test_hip_path_against_recorded_graph_execution(tmp_path, golden_dir, case, capsys):
    try:
        sp, lg = G.export_case(str(tmp_path), case)
    except X.ExporterUnavailable as e:                         # pragma: no cover
        pytest.skip(str(e))
    rc = P.main(G.harness_args(case, sp, lg) + ["--backend", "replay", "--replay", os.path.join(golden_dir, f"onnx_{case}.npz"), "--gpu"])
    out = capsys.readouterr().out
    assert rc == 0, out
    hip = [l for l in out.splitlines() if " hip:" in l]
    assert len(hip) == 3 and all("same_set=True" in l and "order_ok=True" in l for l in hip[:2]) and "identical=True" in hip[2], out


@pytest.mark.gpu
def test_hip_path_against_live_graph_execution_at_the_benchmark_size(tmp_path, capsys):
    """no fixture in between: on the GPU box the exported graph files are executed by tools/mini_onnx.py (torch-CPU) at 480 x 640, K = 1024 (top-k cut active),
    two frames of bench.py's stream, calibrated LightGlue weights -- and librover_fe.so, loaded from the SAME .onnx files through rfe_load_weights' C++ reader
    (--gpu-onnx), must give the graph's keypoints, descriptors <= 1e-4, matches0 and mscores0 <= 1e-4"""
    try:
        sp, lg = G.export_case(str(tmp_path), "s0")
    except X.ExporterUnavailable as e:                         # pragma: no cover
        pytest.skip(str(e))
    rc = P.main(["--superpoint", sp, "--lightglue", lg, "--backend", "mini", "--gpu", "--gpu-onnx", "--frames", "2", "--height", "480", "--width", "640",
                 "--shift-step", "8", "--mscore-tol", "1e-4", "--min-matches", "100"])
    out = capsys.readouterr().out
    assert rc == 0, out
    hip = [l for l in out.splitlines() if " hip:" in l]
    assert len(hip) == 3 and all("K_ref=1024 K=1024 same_set=True" in l and "order_ok=True" in l for l in hip[:2]) and "identical=True" in hip[2], out
