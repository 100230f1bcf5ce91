"""CPU: the `.onnx` reader of the C ABI (rover-slam_amd/csrc/onnx_load.hip -- what rfe_load_weights runs on the paths the reference passes,
src/Extractors/SPextractor.cc:92-94, src/Matchers/lightglue_onnx.cpp:38) under AddressSanitizer + UBSan on a HOST build, against damaged
and hostile files: truncations, byte flips, insertions and -- the class ADVICE r05 reproduced -- length fields that lie (10-byte varints
near 2^64 that wrap `cursor + length`, lengths past the end of the buffer, dims / payload disagreements).  The reader must convert or refuse
with a reason; it must never hang, read outside the file buffer, or let a C++ exception cross the extern "C" boundary."""
import os
import shutil
import subprocess

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from rover_slam_amd import weights as Wt  # noqa: E402
import onnx_export as X  # noqa: E402
import test_onnx_weights as TW  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("fuzz") / "onnx_fuzz_driver"
    cmd = [HIPCC, "--offload-host-only", "-x", "hip", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", os.path.join(ROOT, "rover-slam_amd", "csrc", "onnx_load.hip"), os.path.join(ROOT, "tests", "cpp", "onnx_fuzz_driver.cpp"),
           "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(out)


def _vi(b, i):
    r, s = 0, 0
    while True:
        c = b[i]; i += 1
        r |= (c & 0x7F) << s; s += 7
        if not c & 0x80:
            return r, i


def _fields(b, lo, hi):
    """(field number, wire type, tag offset, payload offset, payload length) of one message's fields"""
    i = lo
    while i < hi:
        t0 = i
        key, i = _vi(b, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            _, j = _vi(b, i); yield fn, wt, t0, i, j - i; i = j
        elif wt == 1:
            yield fn, wt, t0, i, 8; i += 8
        elif wt == 5:
            yield fn, wt, t0, i, 4; i += 4
        else:
            ln, j = _vi(b, i); yield fn, wt, t0, j, ln; i = j + ln


def structure(data):
    """offsets of every length-delimited field header in ModelProto.graph / its initializers / nodes / attributes: (offset of the length varint, what)"""
    spots = []
    for fn, wt, t0, p, ln in _fields(data, 0, len(data)):
        if fn == 7 and wt == 2:
            spots.append((t0 + 1, "graph"))
            for g, gw, gt, gp, gl in _fields(data, p, p + ln):
                if gw != 2:
                    continue
                spots.append((gt + 1, f"graph.{g}"))
                if g in (1, 5) and gl < (1 << 26):
                    for q, qw, qt, qp, ql in _fields(data, gp, gp + gl):
                        if qw == 2:
                            spots.append((qt + 1, f"graph.{g}.{q}"))
                            if g == 1 and q == 5:               # node attribute: one level further (tensor-valued attributes)
                                for r_, rw, rt, rp, rl in _fields(data, qp, qp + ql):
                                    if rw == 2:
                                        spots.append((rt + 1, f"graph.1.5.{r_}"))
                        elif g == 5 and q == 1:                 # TensorProto.dims as unpacked varints
                            spots.append((qp, "dims"))
    return spots


def script_for(data, rng, n_random, n_struct):
    lines = []
    n = len(data)
    spots = structure(data)
    huge = [2 ** 64 - 11, 2 ** 64 - 1, 2 ** 63, 2 ** 63 - 1, 2 ** 32, n + 1, n * 2, 2 ** 40 + 7]
    for k in range(n_struct):                                   # length fields that lie, at real length positions
        off, _ = spots[int(rng.integers(0, len(spots)))]
        v = huge[k % len(huge)]
        lines.append(f"{'L' if v >= 2 ** 56 else 'V'} {off} {v}")
    for k in range(n_struct // 2):                              # small lies: off by a few bytes either way
        off, _ = spots[int(rng.integers(0, len(spots)))]
        lines.append(f"F {off} {int(rng.integers(0, 256))}")
    for k in range(n_random):
        m = k % 4
        if m == 0:
            lines.append(f"T {int(rng.integers(0, n))}")
        elif m == 1:
            lines.append(f"F {int(rng.integers(0, n))} {int(rng.integers(0, 256))}")
        elif m == 2:
            lines.append(f"I {int(rng.integers(0, n))} {int(rng.integers(1, 64))} {k}")
        else:
            lines.append(f"L {int(rng.integers(0, min(n, 1 << 16)))} {huge[k % 3]}")
    return lines


def run(driver, seed, kind, lines, tmp_path):
    sc = tmp_path / "script.txt"
    sc.write_text("\n".join(lines) + "\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([driver, str(seed), str(kind), str(sc), str(tmp_path / "scratch.onnx")], capture_output=True, text=True, env=env, timeout=600)
    tail = r.stdout.strip().splitlines()[-3:]
    assert r.returncode == 0, f"driver exit {r.returncode}\nlast lines: {tail}\nscript line of the failing case: " \
                              f"{lines[len(r.stdout.strip().splitlines())] if len(r.stdout.strip().splitlines()) < len(lines) else '?'}\n{r.stderr[-3000:]}"
    last = r.stdout.strip().splitlines()[-1].split()
    assert last[0] == "done" and int(last[1]) == len(lines)
    return int(last[2]), [tuple(int(x) for x in l.split()) for l in r.stdout.strip().splitlines()[:-1]]


def test_advisor_reproducers(driver, tmp_path):
    """raw_data with a length of 2^64 - 11 in a 30-byte file spun forever; a LayerNormalization whose beta is not an initializer was a null
    dereference; a `posenc.Wr.weight` of the wrong size was read for 64 floats"""
    ld = TW._ld
    # graph { initializer { dims: 4, data_type: 1, name: "w", raw_data: <length lies> } }
    tensor = b"\x08\x04\x10\x01" + ld(8, b"w") + b"\x4a" + bytes([0xF5, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0x01])
    p1 = tmp_path / "wrap.onnx"
    p1.write_bytes(ld(7, ld(5, tensor)))
    g = np.ones(512, np.float32)
    nodes = [TW._node("LayerNormalization", ["x", "g", "beta_from_a_constant_node"], ["y"])]
    p2 = tmp_path / "ln_beta.onnx"
    p2.write_bytes(TW._model([("g", g)], nodes))
    # beta as a larger initializer (the old copy was sized by the tensor, not the manifest)
    p3 = tmp_path / "ln_beta_big.onnx"
    p3.write_bytes(TW._model([("g", g), ("b", np.ones(4096, np.float32))], [TW._node("LayerNormalization", ["x", "g", "b"], ["y"])]))
    p4 = tmp_path / "wr.onnx"
    p4.write_bytes(TW._model([("posenc.Wr.weight", np.ones((3, 5), np.float32)), ("transformers.0.x", np.ones(2, np.float32))], []))
    refused, rows = run(driver, p1, 2, [f"R {p1}", f"R {p2}", f"R {p3}", f"R {p4}"], tmp_path)
    assert refused == 4 and all(r[1] != 0 and r[2] != 0 for r in rows)
    refused, rows = run(driver, p1, 1, [f"R {p1}"], tmp_path)
    assert refused == 1


def test_superpoint_graph_file_under_asan(driver, tmp_path):
    sp, _ = X.export_sp(tmp_path, X.SETTINGS[0], seed=5)
    data = open(sp, "rb").read()
    lines = script_for(data, np.random.default_rng(1), n_random=80, n_struct=120)
    refused, rows = run(driver, sp, 1, ["F 0 " + str(data[0])] + lines, tmp_path)
    assert rows[0][1] == 0 and rows[0][2] == 0            # the unmutated file converts
    assert refused >= 60


def test_lightglue_anonymous_graph_file_under_asan(driver, tmp_path):
    """the onnx-simplifier-style LightGlue file (Linears by order of use, decomposed LayerNorm): the structure route is where placements trusted tensors"""
    cases = [c for c in TW.anonymous_cases(tmp_path) if c[1] == 2 and c[2]]
    path = cases[0][0]
    data = open(path, "rb").read()
    lines = script_for(data, np.random.default_rng(2), n_random=12, n_struct=40)
    refused, rows = run(driver, path, 2, ["F 0 " + str(data[0])] + lines, tmp_path)
    assert rows[0][2] == 0                                 # weights-only conversion of the unmutated file
    assert refused >= 15
