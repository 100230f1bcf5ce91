"""CPU: tools/apply_integration.py performs the CMake edits of INTEGRATION.md on a checkout.  Checked on a synthetic
CMakeLists.txt written here, and -- when the reference checkout is present (build container only; nothing of it is
stored in the repo) -- on a temporary copy of the real one; there the real text of the reference's constructor and four
MatchingPoints_onnx overloads (src/Matchers/SPmatcher.cc) is also compiled, unchanged, against the drop-in headers."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import apply_integration as AI  # noqa: E402

REF = "/root/reference"

FAKE_CMAKE = """cmake_minimum_required(VERSION 3.5)
project(ORB_SLAM3)
set(CMAKE_CXX_FLAGS "-std=c++14 -Wall")
find_package(OpenCV REQUIRED)
find_package(CUDA REQUIRED)
include_directories(
${PROJECT_SOURCE_DIR}
${PROJECT_SOURCE_DIR}/include
/opt/somewhere/onnxruntime-linux-x64-gpu-1.16.3/include
${CUDA_INCLUDE_DIRS}
${OpenCV_INCLUDE_DIRS}
)
add_library(${PROJECT_NAME} SHARED
src/System.cc
src/Extractors/SPextractor.cc
src/Extractors/superpoint_onnx.cc
src/Matchers/SPmatcher.cc
src/Matchers/lightglue_onnx.cpp
src/Matchers/transform.cpp
include/System.h
include/Extractors/SPextractor.h
include/Matchers/SPmatcher.h
include/Settings.h)
target_link_libraries(${PROJECT_NAME}
${CUDA_LIBRARIES}
${OpenCV_LIBS}
/usr/local/lib/libonnxruntime.so
)
add_executable(mono Examples/mono.cc)
"""


def _fake_checkout(tmp_path):
    co = tmp_path / "checkout"
    for p in AI.DROPPED_SOURCES + AI.KEPT_SOURCES:
        f = co / p
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text('#include "Matchers/SPmatcher.h"\n')
    for p in AI.SHADOWED_HEADERS:                       # stand-ins for the reference headers: same guards, a marker macro
        f = co / p
        f.parent.mkdir(parents=True, exist_ok=True)
        guard = {"SPextractor.h": "SPEXTRACTOR_H", "SPmatcher.h": "ORBMATCHER_H"}.get(f.name)
        body = "#define FAKE_REFERENCE_HEADER 1\nnamespace ORB_SLAM3 { class SPextractor; class SPmatcher; }\n"
        f.write_text(f"#ifndef {guard}\n#define {guard}\n{body}#endif\n" if guard else "#pragma once\n" + body)
    # a header of the checkout's own include/ that pulls the two facades in with QUOTED includes, like include/Tracking.h:39,41
    (co / "include" / "Tracking.h").write_text('#pragma once\n#include "Matchers/SPmatcher.h"\n#include "Extractors/SPextractor.h"\n')
    (co / "CMakeLists.txt").write_text(FAKE_CMAKE)
    return co


def _uncommented(text):
    return "\n".join(l.split("#")[0] for l in text.splitlines())


def _check_result(text):
    live = _uncommented(text)
    assert "onnxruntime" not in live and "CUDA" not in live
    assert "add_definitions(-DRFE_WITH_ROVER_SLAM)" in live
    for p in AI.DROPPED_SOURCES:
        assert p not in live
    assert "src/Matchers/SPmatcher.cc" in live and "librover_fe.so" in live and "rfe/ort_compat" in live
    inc = live[live.index("include_directories("):]
    first = [l.strip() for l in inc.splitlines()[1:] if l.strip()][0]
    assert first == os.path.join(ROOT, "include")                      # drop-in headers come first
    assert live.count("(") == live.count(")")


def test_apply_on_synthetic_checkout(tmp_path):
    co = _fake_checkout(tmp_path)
    assert AI.main([str(co), "--dry-run"]) == 0
    assert (co / "CMakeLists.txt").read_text() == FAKE_CMAKE             # dry run writes nothing
    assert AI.main([str(co)]) == 0
    out = (co / "CMakeLists.txt").read_text()
    _check_result(out)
    assert "src/System.cc" in out and "include/Settings.h)" in out and "add_executable(mono" in out
    assert (co / "CMakeLists.txt.pre_rfe").read_text() == FAKE_CMAKE
    assert AI.main([str(co)]) == 0 and (co / "CMakeLists.txt").read_text() == out   # idempotent
    for p in AI.SHADOWED_HEADERS:                                         # the six headers are forwarders now, originals kept
        assert AI.is_forwarder(str(co / p)) and "FAKE_REFERENCE_HEADER" in (co / (p + ".pre_rfe")).read_text()
    # a forwarder that was put back (e.g. `git checkout include/`) is a failed check, and a re-run repairs it
    shutil.copy(co / (AI.SHADOWED_HEADERS[0] + ".pre_rfe"), co / AI.SHADOWED_HEADERS[0])
    assert any("still is the reference header" in x for x in AI.check_tree(str(co), ROOT, applied=True))
    assert AI.main([str(co)]) == 0 and AI.check_tree(str(co), ROOT, applied=True) == []
    # --revert: everything back
    assert AI.main([str(co), "--revert"]) == 0
    assert (co / "CMakeLists.txt").read_text() == FAKE_CMAKE and not (co / "CMakeLists.txt.pre_rfe").exists()
    for p in AI.SHADOWED_HEADERS:
        assert "FAKE_REFERENCE_HEADER" in (co / p).read_text() and not (co / (p + ".pre_rfe")).exists()
    assert AI.main([str(co)]) == 0
    # a file that is not Rover-SLAM's is refused, untouched
    (co / "CMakeLists.txt").write_text("project(x)\nadd_library(x a.cc)\n")
    assert AI.main([str(co)]) == 3 and (co / "CMakeLists.txt").read_text() == "project(x)\nadd_library(x a.cc)\n"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "CMakeLists.txt")), reason="reference checkout not present (GPU box)")
def test_apply_on_copy_of_reference(tmp_path):
    co = tmp_path / "ref"
    (co / "src").mkdir(parents=True)
    shutil.copy(os.path.join(REF, "CMakeLists.txt"), co / "CMakeLists.txt")
    for d in ("src/Extractors", "src/Matchers", "include/Matchers", "include/Extractors"):
        shutil.copytree(os.path.join(REF, d), co / d)
    assert AI.main([str(co)]) == 0
    _check_result((co / "CMakeLists.txt").read_text())
    for p in AI.SHADOWED_HEADERS:                      # the real headers: renamed byte for byte, forwarders in their place
        assert AI.is_forwarder(str(co / p))
        assert (co / (p + ".pre_rfe")).read_bytes() == open(os.path.join(REF, p), "rb").read()
    before = open(os.path.join(REF, "CMakeLists.txt")).read().splitlines()
    after = (co / "CMakeLists.txt").read_text().splitlines()
    assert abs(len(after) - len(before)) < 20                            # a handful of lines change, nothing else


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_quoted_include_from_checkout_header_sees_the_dropin(tmp_path):
    """Tracking.h / LocalMapping.h / LoopClosing.h include "Matchers/SPmatcher.h" / "Extractors/SPextractor.h"
    with QUOTES, and a quoted include searches the including file's directory before any -I path -- so -I<this repo>/include
    first does NOT shadow the reference headers for them.  A translation unit that includes a header located in the checkout's
    include/ must see the drop-in classes after apply_integration (forwarders), and demonstrably did not before."""
    co = _fake_checkout(tmp_path)
    tu = tmp_path / "tu.cc"
    tu.write_text('#include "Tracking.h"\n'
                  "#ifdef FAKE_REFERENCE_HEADER\n#error the reference header won over the drop-in\n#endif\n"
                  "static_assert(sizeof(ORB_SLAM3::SPextractor) > 0 && sizeof(ORB_SLAM3::SPmatcher) > 0, \"drop-in classes are complete types\");\n"
                  "int main() { return rfe_version() == nullptr; }\n")
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-DRFE_NO_OPENCV", "-I" + os.path.join(ROOT, "include"), "-I" + str(co / "include"), str(tu)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode != 0 and "the reference header won" in r.stderr          # the defect, before the forwarders exist
    assert AI.main([str(co)]) == 0
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def _function_bodies(src, head_regex):
    """Source text of every function definition whose head matches, by brace matching."""
    out = []
    for m in re.finditer(head_regex, src):
        i = src.index("{", m.end())
        depth, j = 1, i + 1
        while depth:
            depth += {"{": 1, "}": -1}.get(src[j], 0)
            j += 1
        out.append(src[m.start():j])
    return out


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src/Matchers/SPmatcher.cc")) or shutil.which("g++") is None,
                    reason="reference checkout not present (GPU box) or no g++")
def test_reference_spmatcher_runner_code_compiles_unchanged(tmp_path):
    """The reference's own text -- include block, the three constants, the constructor and the four MatchingPoints_onnx
    overloads of src/Matchers/SPmatcher.cc, which is every line of that file that touches the runner -- compiled as it is
    (-std=c++14 like CMakeLists.txt:12) against include/Matchers/SPmatcher.h in -DRFE_WITH_ROVER_SLAM mode, with
    rfe/ort_compat standing in for onnxruntime and test stand-ins for the headers this image lacks (OpenCV, Eigen, Sophus,
    Frame/KeyFrame/MapPoint).  The extract lives in tmp_path only."""
    src = open(os.path.join(REF, "src/Matchers/SPmatcher.cc"), errors="replace").read()
    head = src[:src.index("SPmatcher::SPmatcher(")]
    assert "onnxruntime_cxx_api.h" in head and "SPmatcher::TH_HIGH" in head
    ctor = _function_bodies(src, r"SPmatcher::SPmatcher\(float thre\)")
    over = _function_bodies(src, r"\nint SPmatcher::MatchingPoints_onnx\(")
    assert len(ctor) == 1 and len(over) == 4 and all("Ort::Value" in o for o in over)
    tu = tmp_path / "spmatcher_runner_slice.cc"
    tu.write_text(head + ctor[0] + "\n" + "\n".join(over) + "\n}\n")
    extra = tmp_path / "inc" / "opencv2" / "core"
    extra.mkdir(parents=True)
    (tmp_path / "inc" / "opencv2" / "core.hpp").write_text('#pragma once\n#include "opencv2/core/core.hpp"\n')
    (extra / "eigen.hpp").write_text("#pragma once\n")
    stubs = os.path.join(ROOT, "tests", "cpp", "rover_slam_stubs")
    cmd = ["g++", "-std=c++14", "-fopenmp", "-pthread", "-DRFE_WITH_ROVER_SLAM", "-DRFE_NO_OPENCV", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "include", "rfe", "ort_compat"), "-I" + stubs, "-I" + str(tmp_path / "inc"), "-c", str(tu), "-o",
           str(tmp_path / "slice.o")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    nm = subprocess.run(["nm", "-C", str(tmp_path / "slice.o")], capture_output=True, text=True).stdout
    assert nm.count("ORB_SLAM3::SPmatcher::MatchingPoints_onnx(") == 4 and "ORB_SLAM3::SPmatcher::SPmatcher(float)" in nm
