#!/usr/bin/env python3
"""Wire librover_fe.so into a Rover-SLAM checkout: the CMake edits of INTEGRATION.md section 1, performed.

    python tools/apply_integration.py /path/to/Rover-SLAM [--rfe-root /path/to/this/repo] [--dry-run]

What changes: CMakeLists.txt, and SIX reference headers are renamed and replaced by one-line forwarders (no .cc file is edited,
nothing of the reference is copied anywhere):
  six headers           include/Extractors/{SPextractor,superpoint_onnx}.h and include/Matchers/{SPmatcher,lightglue_onnx,
                        Configuration,transform}.h of the CHECKOUT are renamed to <name>.pre_rfe and replaced by a forwarder
                        `#include "<this repo>/include/<same path>"`.  Putting this repo's include/ first on the -I list is not enough:
                        the reference's own headers pull them in with quoted includes (Tracking.h:39,41, LocalMapping.h:29,
                        LoopClosing.h:35), and a quoted include searches the INCLUDING file's directory -- the checkout's include/ --
                        before any -I path, so Tracking.cc / System.cc / LocalMapping.cc / LoopClosing.cc would still see the
                        reference classes (whose out-of-line bodies were just dropped from add_library).
  include_directories   this repo's include/ goes FIRST (angle-bracket / src-relative includes resolve to the drop-in headers too);
                        the onnxruntime include directory (CMakeLists.txt:63) is replaced by
                        include/rfe/ort_compat (an `Ort::Value` = rfe::Tensor alias: src/Matchers/SPmatcher.cc compiles unchanged);
                        ${CUDA_INCLUDE_DIRS} goes.
  find_package(CUDA)    commented out (:50).
  add_definitions       -DRFE_WITH_ROVER_SLAM: include/Matchers/SPmatcher.h then only declares the reference's full member list
                        and every body stays in the reference's src/Matchers/SPmatcher.cc.
  add_library sources   the four sources whose classes / functions are header-only now are dropped: src/Extractors/SPextractor.cc,
                        src/Extractors/superpoint_onnx.cc, src/Matchers/lightglue_onnx.cpp, src/Matchers/transform.cpp (:106-110),
                        and the shadowed reference headers leave the listing (:142-147).  src/Matchers/SPmatcher.cc STAYS.
  target_link_libraries /usr/local/lib/libonnxruntime.so (:161) -> rover-slam_amd/librover_fe.so; ${CUDA_LIBRARIES} goes.
The original is kept as CMakeLists.txt.pre_rfe.  Idempotent: a second run reports "already applied" (and still repairs missing
forwarders).  --revert restores CMakeLists.txt and the six headers from their .pre_rfe copies.
"""
import argparse
import os
import re
import shutil
import sys

DROPPED_SOURCES = ["src/Extractors/SPextractor.cc", "src/Extractors/superpoint_onnx.cc", "src/Matchers/lightglue_onnx.cpp",
                   "src/Matchers/transform.cpp"]
SHADOWED_HEADERS = ["include/Extractors/SPextractor.h", "include/Extractors/superpoint_onnx.h", "include/Matchers/SPmatcher.h",
                    "include/Matchers/lightglue_onnx.h", "include/Matchers/Configuration.h", "include/Matchers/transform.h"]
KEPT_SOURCES = ["src/Matchers/SPmatcher.cc"]
MARK = "# --- rover_fe (librover_fe.so) integration, tools/apply_integration.py ---"
FWD_MARK = "// rover_fe forwarder (tools/apply_integration.py)"


class IntegrationError(RuntimeError):
    pass


def transform_cmake(text, rfe_root):
    """Returns (new_text, report lines).  Raises IntegrationError when the file does not look like Rover-SLAM's."""
    if MARK in text:
        return text, ["already applied"]
    rep = []
    inc, lib = os.path.join(rfe_root, "include"), os.path.join(rfe_root, "rover-slam_amd", "librover_fe.so")
    lines = text.splitlines()
    out, block = [], None
    seen = {"include_block": False, "ort_include": False, "ort_lib": False, "sources": set(), "kept": set()}
    for ln in lines:
        s = ln.strip()
        if block is None:
            if re.match(r"include_directories\s*\(", s):
                block = "inc"; seen["include_block"] = True
                out.append(ln)
                out.append(inc + "   # rover_fe drop-in headers shadow the reference's same-named ones")
                rep.append(f"include_directories: + {inc} (first)")
                continue
            if re.match(r"add_library\s*\(\s*\$\{PROJECT_NAME\}", s):
                block = "lib"
            elif re.match(r"target_link_libraries\s*\(\s*\$\{PROJECT_NAME\}", s):
                block = "link"
            elif re.match(r"find_package\s*\(\s*CUDA\b", s):
                out.append("# " + ln + "   # rover_fe: no CUDA")
                rep.append("find_package(CUDA): commented out")
                continue
            out.append(ln)
            continue
        # inside a block
        if block == "inc":
            if "onnxruntime" in s:
                out.append(os.path.join(inc, "rfe", "ort_compat") + "   # was: " + s)
                seen["ort_include"] = True
                rep.append(f"include_directories: {s} -> rfe/ort_compat")
            elif s == "${CUDA_INCLUDE_DIRS}":
                rep.append("include_directories: - ${CUDA_INCLUDE_DIRS}")
            else:
                out.append(ln)
        elif block == "lib":
            bare = s.rstrip(")").strip()
            if bare in DROPPED_SOURCES or bare in SHADOWED_HEADERS:
                seen["sources"].add(bare)
                rep.append(f"add_library: - {bare}")
                if s.endswith(")"):
                    out.append(")")
            else:
                if bare in KEPT_SOURCES:
                    seen["kept"].add(bare)
                out.append(ln)
        elif block == "link":
            if "onnxruntime" in s:
                out.append(lib + ("\n)" if s.endswith(")") else ""))
                seen["ort_lib"] = True
                rep.append(f"target_link_libraries: {s.rstrip(')')} -> {lib}")
            elif s == "${CUDA_LIBRARIES}":
                rep.append("target_link_libraries: - ${CUDA_LIBRARIES}")
            else:
                out.append(ln)
        if s.endswith(")"):
            block = None
    missing = [k for k in ("include_block", "ort_include", "ort_lib") if not seen[k]]
    missing += [f"source {p}" for p in DROPPED_SOURCES if p not in seen["sources"]]
    missing += [f"kept source {p}" for p in KEPT_SOURCES if p not in seen["kept"]]
    if missing:
        raise IntegrationError("CMakeLists.txt does not look like Rover-SLAM's (not found: " + ", ".join(missing) + ")")
    # the definition goes right after project(...)
    for i, ln in enumerate(out):
        if re.match(r"\s*project\s*\(", ln):
            out[i + 1:i + 1] = ["", MARK, "add_definitions(-DRFE_WITH_ROVER_SLAM)"]
            rep.append("add_definitions(-DRFE_WITH_ROVER_SLAM)")
            break
    else:
        raise IntegrationError("no project() line")
    return "\n".join(out) + "\n", rep


def is_forwarder(path):
    try:
        with open(path, errors="replace") as f:
            return f.read(200).startswith(FWD_MARK)
    except OSError:
        return False


def install_forwarders(checkout, rfe_root, dry_run=False):
    """Rename the six shadowed reference headers to <name>.pre_rfe and put a one-line forwarder to the drop-in header in their
    place.  Returns report lines."""
    rep = []
    for rel in SHADOWED_HEADERS:
        dst = os.path.join(checkout, rel)
        if is_forwarder(dst):
            continue
        target = os.path.join(os.path.abspath(rfe_root), rel)
        rep.append(f"header: {rel} -> {rel}.pre_rfe, forwarder to {target}")
        if dry_run:
            continue
        if os.path.exists(dst):
            os.replace(dst, dst + ".pre_rfe")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w") as f:
            f.write(f"{FWD_MARK}: the reference header is kept beside this file as {os.path.basename(rel)}.pre_rfe\n"
                    f'#include "{target}"\n')
    return rep


def revert(checkout):
    """Undo apply: CMakeLists.txt and the six headers come back from their .pre_rfe copies."""
    rep = []
    for rel in ["CMakeLists.txt"] + SHADOWED_HEADERS:
        p = os.path.join(checkout, rel)
        if os.path.exists(p + ".pre_rfe"):
            os.replace(p + ".pre_rfe", p)
            rep.append(f"restored {rel}")
        elif rel != "CMakeLists.txt" and is_forwarder(p):
            os.remove(p)
            rep.append(f"removed forwarder {rel} (no .pre_rfe copy)")
    return rep


def check_tree(checkout, rfe_root, applied=False):
    """The things the edit relies on, checked on the actual trees.  applied=True (after the edit): every shadowed header of the
    checkout must be a forwarder -- a reference header left in place would win over the drop-in for every quoted include
    that comes from the checkout's own include/ directory."""
    problems = []
    for p in DROPPED_SOURCES + KEPT_SOURCES:
        if not os.path.exists(os.path.join(checkout, p)):
            problems.append(f"checkout has no {p}")
    for p in SHADOWED_HEADERS:
        if not os.path.exists(os.path.join(rfe_root, p)):
            problems.append(f"drop-in header missing: {p}")
        if applied and not is_forwarder(os.path.join(checkout, p)):
            problems.append(f"{p} of the checkout still is the reference header (it shadows the drop-in for quoted includes)")
    for p in ("include/rfe/ort_compat/onnxruntime_cxx_api.h", "include/rover_fe.h"):
        if not os.path.exists(os.path.join(rfe_root, p)):
            problems.append(f"missing {p}")
    # every quoted include of the kept source must resolve to a drop-in header or to something the checkout still has
    src = open(os.path.join(checkout, KEPT_SOURCES[0]), errors="replace").read() if not problems else ""
    for inc in re.findall(r'#include\s*"([^"]+)"', src):
        if not (os.path.exists(os.path.join(rfe_root, "include", inc)) or os.path.exists(os.path.join(checkout, "include", inc))
                or os.path.exists(os.path.join(checkout, inc)) or inc in ("iostream",)):
            problems.append(f"{KEPT_SOURCES[0]} includes \"{inc}\", found in neither tree")
    return problems


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("checkout")
    ap.add_argument("--rfe-root", default=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    ap.add_argument("--dry-run", action="store_true", help="print the report, write nothing")
    ap.add_argument("--revert", action="store_true", help="restore CMakeLists.txt and the six headers from their .pre_rfe copies")
    a = ap.parse_args(argv)
    cm = os.path.join(a.checkout, "CMakeLists.txt")
    if not os.path.exists(cm):
        print(f"apply_integration: {cm} not found", file=sys.stderr)
        return 2
    if a.revert:
        for r in revert(a.checkout) or ["nothing to revert"]:
            print(r)
        return 0
    problems = check_tree(a.checkout, a.rfe_root)
    if problems:
        print("apply_integration: " + "; ".join(problems), file=sys.stderr)
        return 3
    try:
        new, rep = transform_cmake(open(cm).read(), os.path.abspath(a.rfe_root))
    except IntegrationError as e:
        print(f"apply_integration: {e}", file=sys.stderr)
        return 3
    already = rep == ["already applied"]
    rep += install_forwarders(a.checkout, a.rfe_root, dry_run=a.dry_run)
    for r in rep:
        print(r)
    if a.dry_run:
        return 0
    if not already:
        if not os.path.exists(os.path.join(a.rfe_root, "rover-slam_amd", "librover_fe.so")):
            print("note: rover-slam_amd/librover_fe.so is not built yet (make -C rover-slam_amd/csrc)")
        shutil.copy2(cm, cm + ".pre_rfe")
        with open(cm, "w") as f:
            f.write(new)
        print(f"wrote {cm} (original kept as CMakeLists.txt.pre_rfe); model files: the reference's own onnxmodel/superpoint.onnx, onnxmodel/lightglue_sim.onnx (read by librover_fe.so itself; RFEW containers work too) "
              "(python -m rover_slam_amd.onnx_weights converts the .onnx initialisers)")
    problems = check_tree(a.checkout, a.rfe_root, applied=True)
    if problems:
        print("apply_integration: " + "; ".join(problems), file=sys.stderr)
        return 3
    return 0


if __name__ == "__main__":
    sys.exit(main())
