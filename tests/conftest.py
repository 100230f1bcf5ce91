import os
import sys

# The oracle (oracle/librfe_oracle.so) is OpenMP code: with libgomp's default ACTIVE wait policy its threads spin at every barrier, and on a host whose cores
# are busy with something else (the round-6 "noisy host" run: a busy loop on every core) each barrier then waits for descheduled spinners -- the GPU suite
# went from 3.4 to > 25 minutes.  Blocking waits keep the checker usable there; set before libgomp is first loaded.  (bench.py's cpu_baseline leg, which TIMES
# the oracle, does not go through this file and keeps the default policy.)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))   # tests/tolerances.py
if os.path.join(ROOT, "tools") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tools"))                  # tools/onnx_export.py, tools/mini_onnx.py (build-container infrastructure)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# Collection order under `-x`: the tests that compare the HIP path with oracle/ or tests/golden/ run FIRST, kernel-vs-float64 formula
# files next, and the bench.py subprocess tests (the most environment-sensitive code of the repo: child processes, process groups,
# wall clocks) LAST -- a hiccup there must never keep a parity test from running.  Files not named keep their alphabetical place in the middle.
_ORDER_FIRST = ["test_gpu_parity", "test_gpu_calibrated", "test_gpu_throughput_parity", "test_stereo", "test_gpu_shim", "test_gpu_onnx_graph",
                "test_onnx_cpp", "test_onnx_weights", "test_onnx_hparams", "test_onnx_exporter", "test_ort_parity"]
# test_pool.py: oracle-checked too, but on a box with SEVERAL GPUs it runs RCCL between distinct devices -- code no box of this build has ever executed --
# so it goes behind every single-GPU file and in front of the bench subprocess tests only
_ORDER_LAST = ["test_pool", "test_gpu_bench"]


def _rank(item):
    mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if mod in _ORDER_FIRST:
        return (0, _ORDER_FIRST.index(mod))
    if mod in _ORDER_LAST:
        return (2, _ORDER_LAST.index(mod))
    return (1, 0)


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=_rank)      # stable: the order inside a file, and among unnamed files, is unchanged
