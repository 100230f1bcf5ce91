"""CPU: the C-ABI library loads and exports every symbol include/rover_fe.h declares; the drop-in C++
headers compile and link against it; without a GPU every entry point fails loudly (no CPU fallback)."""
import ctypes
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rover_fe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rfe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from rover_slam_amd import capi
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(capi.lib, s), f"librover_fe.so does not export {s}"
    assert set(capi.EXPORTS) == set(syms)
    assert capi.lib.rfe_weight_count(1) == 1300865 and capi.lib.rfe_weight_count(2) == 11321153
    assert b"gfx950" in capi.lib.rfe_version()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    from rover_slam_amd import capi
    with pytest.raises(capi.RfeError) as e:
        capi.Context(0)
    assert "no HIP device" in str(e.value) or "hip" in str(e.value).lower()
    # NULL ctx is rejected, not dereferenced
    assert capi.lib.rfe_synchronize(None) < 0


def test_weight_container_roundtrip(tmp_path):
    from rover_slam_amd import weights as Wt
    blob = Wt.make_superpoint(seed=3)
    p = str(tmp_path / "sp.rfew")
    Wt.save(p, blob, 1)
    back, kind = Wt.load(p)
    assert kind == 1 and np.array_equal(back, blob)
    assert Wt.SP_COUNT == 1300865 and Wt.LG_COUNT == 11321153
    # the manifest is the canonical layout: contiguous, complete
    man, n = Wt.lg_manifest()
    off = 0
    for _, o, shape in man:
        assert o == off
        off += int(np.prod(shape))
    assert off == n


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_shim_headers_compile_and_link(tmp_path):
    """include/Extractors/SPextractor.h, include/Matchers/SPmatcher.h, include/SuperPoint.h and
    include/super_glue.h build without OpenCV / onnxruntime and link against librover_fe.so."""
    exe = str(tmp_path / "shim_driver")
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),   # the reference builds -std=c++14 (CMakeLists.txt:12)
           os.path.join(ROOT, "tests", "cpp", "shim_driver.cpp"), "-o", exe,
           "-L" + os.path.join(ROOT, "rover-slam_amd"), "-lrover_fe", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.join(ROOT, "rover-slam_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(exe)


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_spmatcher_in_tree_mode_cxx14(tmp_path):
    """-DRFE_WITH_ROVER_SLAM: include/Matchers/SPmatcher.h only DECLARES (full member list of the reference class) and a
    reference-style SPmatcher.cc -- out-of-line constants, constructor and MatchingPoints_onnx bodies written against
    Ort::Value (rfe/ort_compat) -- compiles against it with -std=c++14 -Wall -Werror; the constants are defined exactly
    once (no C++17 inline variables in the header) and the object links against librover_fe.so."""
    stubs = os.path.join(ROOT, "tests", "cpp", "rover_slam_stubs")
    obj = str(tmp_path / "bodies.o")
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-DRFE_WITH_ROVER_SLAM", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "include", "rfe", "ort_compat"), "-I" + stubs, "-c", os.path.join(stubs, "onnx_matcher_bodies.cc"), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    nm = subprocess.run(["nm", "-C", obj], capture_output=True, text=True).stdout
    for sym in ("ORB_SLAM3::SPmatcher::TH_HIGH", "ORB_SLAM3::SPmatcher::TH_LOW", "ORB_SLAM3::SPmatcher::HISTO_LENGTH"):
        defs = [l for l in nm.splitlines() if l.endswith(sym) and l.split()[-2] in "RDrdB"]
        assert len(defs) == 1, (sym, nm)
    main = tmp_path / "main.cc"
    main.write_text("int rfe_in_tree_probe();\nint main() { return rfe_in_tree_probe() == 31 ? 0 : 1; }\n")
    exe = str(tmp_path / "probe")
    r = subprocess.run(["g++", "-std=c++14", str(main), obj, "-o", exe, "-L" + os.path.join(ROOT, "rover-slam_amd"), "-lrover_fe",
                        "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "rover-slam_amd"), "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the stand-alone header in a second translation unit next to the first: no duplicate definitions in C++14 either
    for i in (1, 2):
        (tmp_path / f"tu{i}.cc").write_text('#include "Matchers/SPmatcher.h"\nfloat tu%d() { return ORB_SLAM3::SPmatcher::TH_HIGH; }\n' % i)
    (tmp_path / "tumain.cc").write_text("float tu1(); float tu2(); int main() { return tu1() == tu2() ? 0 : 1; }\n")
    r = subprocess.run(["g++", "-std=c++14", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(tmp_path / "tu1.cc"), str(tmp_path / "tu2.cc"),
                        str(tmp_path / "tumain.cc"), "-o", str(tmp_path / "tu"), "-L" + os.path.join(ROOT, "rover-slam_amd"), "-lrover_fe",
                        "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "rover-slam_amd"), "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
