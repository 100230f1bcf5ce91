"""GPU: bench.py's own contract on the box the driver uses -- one JSON line with the fields the task statement names, and
the N-rank flow (bench.py --gpus 2 spawning two ranks that share the one GPU over gloo: a functional check of sharding,
rank census and the ONE-gather result path; RCCL itself needs N GPUs and is exercised by the driver's scaling run)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:      # the reason lives in the JSON line on stdout (invalid / invalid_aux / *.error), not only on stderr
        why = {}
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            why = {k: d[k] for k in ("invalid", "invalid_aux", "perf_notes") if k in d}
            for sect in ("variants", "latency", "pool_c_abi", "pcie_inclusive"):
                v = d.get(sect)
                if isinstance(v, dict):
                    errs = {k: x["error"] for k, x in v.items() if isinstance(x, dict) and "error" in x}
                    if "error" in v and isinstance(v["error"], str):
                        errs["error"] = v["error"]
                    if errs:
                        why[sect + ".errors"] = errs
        except (ValueError, IndexError):
            why = {"stdout_tail": r.stdout[-1500:]}
        pytest.fail(f"bench.py {' '.join(argv)} exited {r.returncode}\nline says: {json.dumps(why, indent=1)[:3000]}\nstderr tail:\n{r.stderr[-2000:]}")
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_line_contract_single_gpu():
    d = _bench("--steps", "3", "--warmup", "1", "--sustained-steps", "20", "--no-cpu-baseline")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "sustained"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["scaling"] == "weak" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0.3 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3          # 32 frames per step per GPU
    s = d["sustained"]
    assert s["steps"] == 20 and s["ms_per_step"]["min"] <= s["ms_per_step"]["mean"] <= s["ms_per_step"]["max"]


def test_bench_gpus2_over_gloo_on_one_gpu():
    d = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--sustained-steps", "0", env={"RFE_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and "child processes" in d["launcher"]
    assert d["rccl"]["world_size"] == 2 and d["rccl"]["allreduce_sum_of_ones"] == 2 and len(d["rccl"]["devices"]) == 2
    assert d["gather"]["collectives_per_step"] == 1 and d["gather"]["last_step_payload_verified"] is True
    assert len(d["per_rank_ms_per_step"]["all"]) == 2 and isinstance(d["per_rank_ms_per_step"]["outlier"], bool)
    assert d["rccl"]["untimed_collective_warmup_rounds"] >= 3          # whatever --warmup says, the timed region never holds the group's first collectives
    assert abs(d["value"] - 2 * 32 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3      # whole-job frames over the max-over-ranks time


def test_bench_rccl_code_path_at_world_size_one():
    """RCCL needs one GPU per rank, so the N-rank run belongs to the driver's 8-GPU box; what CAN run here is the same code path
    at world size 1 (RFE_BENCH_FORCE_PG=1): process group on the "nccl" backend bound to the device, object all-gather, device
    all-reduce, barrier-bracketed timing and the per-step dist.gather of the packed results into the preallocated buffer."""
    d = _bench("--steps", "2", "--warmup", "1", "--sustained-steps", "4", "--no-cpu-baseline", "--no-pcie", env={"RFE_BENCH_FORCE_PG": "1"})
    assert d["n_gpus"] == 1 and d["rccl"]["backend"] == "nccl" and d["rccl"]["world_size"] == 1 and d["rccl"]["allreduce_sum_of_ones"] == 1
    assert d["gather"]["collectives_per_step"] == 1 and d["gather"]["last_step_payload_verified"] is True
    assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3
    assert d["rccl"]["untimed_collective_warmup_rounds"] >= 3          # the timed region never holds a communicator's first collectives
    assert "r04_pairing_step_time_ratio" not in d.get("perf_notes", {})  # no step-time comparison against a loop that holds collectives
