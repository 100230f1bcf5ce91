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
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "shim_driver.cpp"), "-o", exe,
           "-L" + os.path.join(ROOT, "rover-slam_amd"), "-lrover_fe", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.join(ROOT, "rover-slam_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(exe)
