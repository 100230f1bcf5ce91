"""CPU: bench.py's exit-status plumbing.  Only RESULT mismatches (the timed loop's outputs or an auxiliary run's, checked against the oracle
or against the resident path) make the exit status non-zero; timings and infrastructure failures never do."""
import copy
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

GOOD = {
    "value": 937.3,
    "cpu_baseline": {"verified_against_gpu": {"ok": True}},
    "variants": {"fp16x2": {"value": 1500.0, "verified_against_oracle": {"ok": True}}, "kmax512": {"value": 2000.0}},
    "latency": {"note": "text", "resident": {"c2": {"ms": 0.6}, "verified_against_oracle": {"ok": True}},
                "dropin": {"c2_ms": 0.7, "verified_against_oracle": {"ok": True, "c3_matches_oracle": 170}},
                "dropin_host_graph": {"verified_against_oracle": {"ok": True}},
                "resident_fp16x2": {"c3": {"ms": 1.5}, "verified_against_oracle": {"ok": True}}},
    "perf_notes": {"r04_pairing_step_time_ratio": {"value": 1.9}},       # a wild timing ratio is information, not a verdict
}


def _line(**patch):
    d = copy.deepcopy(GOOD)
    for path, v in patch.items():
        cur = d
        keys = path.split("__")
        for k in keys[:-1]:
            cur = cur[k]
        cur[keys[-1]] = v
    return d


def test_clean_line_exits_zero_whatever_the_timings_say():
    d = _line()
    assert bench.exit_status(d, []) == 0 and "invalid" not in d and "invalid_aux" not in d and d["value"] == 937.3


def test_every_oracle_checked_auxiliary_run_reaches_the_exit_status():
    for path in ("latency__resident__verified_against_oracle", "latency__dropin__verified_against_oracle",
                 "latency__dropin_host_graph__verified_against_oracle", "latency__resident_fp16x2__verified_against_oracle",
                 "variants__fp16x2__verified_against_oracle"):
        d = _line(**{path: {"ok": False, "c3_match_list_agrees": False}})
        assert bench.exit_status(d, []) == 5, path
        assert len(d["invalid_aux"]) == 1 and path.split("__")[1] in d["invalid_aux"][0]
        assert d["value"] == 937.3          # the headline's own check passed: the value stays, the line is flagged


def test_library_errors_count_infrastructure_errors_do_not():
    d = _line(latency__dropin={"error": "RfeError: rfe_match: invalid argument (-2)"})
    assert bench.exit_status(d, []) == 5 and "latency.dropin" in d["invalid_aux"][0]
    d = _line(latency__dropin={"error": "/x/lat_driver missing (python -c 'import __graft_entry__ as g; g.build()')"})
    assert bench.exit_status(d, []) == 0
    d = _line(latency__resident={"error": "OutOfMemoryError: HIP out of memory"})
    assert bench.exit_status(d, []) == 0
    d = _line(variants__error="RfeError: rfe_extract_match_stream_dev: launch failed")
    assert bench.exit_status(d, []) == 5


def test_headline_mismatch_is_exit_four_and_withholds_the_value():
    d = _line(cpu_baseline__verified_against_gpu={"ok": False})
    assert bench.exit_status(d, []) == 4 and d["value"] is None and d["value_unverified"] == 937.3 and "invalid" in d
    d = _line(cpu_baseline__verified_against_gpu={"ok": False}, latency__dropin__verified_against_oracle={"ok": False})
    assert bench.exit_status(d, ["pool_c_abi: differs"]) == 4 and len(d["invalid_aux"]) == 2


def test_resident_path_mismatches_collected_during_the_run_still_count():
    d = _line()
    assert bench.exit_status(d, ["variants.strong_n1: the first 33 frames differ"]) == 5 and d["invalid_aux"] == ["variants.strong_n1: the first 33 frames differ"]


def test_no_timing_enters_the_verdict():
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def oracle_mismatches"):src.index("class AuxMismatch")]
    for word in ("ms_per_step", "ratio", "perf_counter", "elapsed"):
        assert word not in body, word
    assert "step_time_ratio_to_headline" not in src
