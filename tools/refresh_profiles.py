#!/usr/bin/env python3
"""Turn the outputs of tools/profile_round.sh <tag> (gpurun_out/<tag>/) into the committed summaries:
profiles/r01_bench.json, r01_kernel_stats.md, r01_pmc.md (HBM and SQ tables only; the hand-written notes below them are
kept), r01_pmc_traffic.json.   usage: python tools/refresh_profiles.py r01v9"""
import io
import json
import os
import re
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_pmc
import rocpd_stats


def capture(fn, *a):
    buf = io.StringIO()
    with redirect_stdout(buf):
        fn(*a)
    return buf.getvalue()


def split(t):
    i = t.index("```json")
    return t[:i].strip(), json.loads(t[i + 7:t.rindex("```")])


def main(tag):
    O = os.path.join(ROOT, "gpurun_out", tag)
    P = os.path.join(ROOT, "profiles")
    bench = open(os.path.join(O, "bench.json")).read().strip().splitlines()[-1]
    d = json.loads(bench)
    open(os.path.join(P, "r01_bench.json"), "w").write(bench + "\n")
    ks = capture(rocpd_stats.main, os.path.join(O, "trace", "t_results.db")).replace(ROOT + "/", "")
    att = re.search(r"lg_attention_kernel<false, 0, false>` \| \d+ \| [\d.]+ \| ([\d.]+)", ks)
    head = f"""# Round 1 — rocprofv3 --kernel-trace summary (final round-1 kernels)

Command (MI355X box, tools/profile_round.sh {tag}): `rocprofv3 --kernel-trace --stats -d gpurun_out/{tag}/trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` (1 warm-up + 3 fully instrumented + 3 timed steps of 33 frames + 32 pairs, plus the steps of the PCIe-inclusive loop; rocpd sqlite summarised by tools/rocpd_stats.py).

Un-profiled bench.py line from the same box just before: profiles/r01_bench.json ({d['value']} frames/s, {d['ms_per_step']} ms/step; lg_attention avg {d['roofline']['avg_launch_ms'] * 1e3:.1f} us by HIP events in the timed region, {att.group(1) if att else '?'} us in this trace).

"""
    open(os.path.join(P, "r01_kernel_stats.md"), "w").write(head + ks)
    hbm_tab, hj = split(capture(rocpd_pmc.main, [os.path.join(O, "pmc_fetch", "f_results.db"), os.path.join(O, "pmc_write", "w_results.db")]))
    sq_tab, _ = split(capture(rocpd_pmc.main, [os.path.join(O, "pmc_sq", "s_results.db")]))
    out = {"note": "per-launch HBM traffic from rocprofv3 PMC, (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes, bench.py workload (33 frames / 32 pairs "
                   f"per step), final round-1 kernels; source gpurun_out/{tag} (tools/profile_round.sh)",
           "kernels": {k: {"fetch_kib": v.get("FETCH_SIZE"), "write_kib": v.get("WRITE_SIZE"), "traffic_bytes": v.get("traffic_bytes"),
                           "avg_us": v.get("avg_us"), "calls": v.get("calls")} for k, v in hj.items()}}
    json.dump(out, open(os.path.join(P, "r01_pmc_traffic.json"), "w"), indent=1)
    p = os.path.join(P, "r01_pmc.md")
    s = open(p).read()
    a, b = s.index("## HBM"), s.index("## Instruction mix")
    s = s[:a] + "## HBM\n" + hbm_tab + "\n\n## SQ / GRBM\n" + sq_tab + "\n\n" + s[b:]
    s = re.sub(r"tools/profile_round.sh r01v\d+", f"tools/profile_round.sh {tag}", s)
    open(p, "w").write(s)
    print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d.get("pcie_inclusive", {}).get("value"),
          d.get("cpu_baseline", {}).get("value"))
    for w in ("c2", "c3", "c5"):
        f = os.path.join(O, f"lat_{w}.json")
        if os.path.exists(f):
            x = json.loads(open(f).read().strip().splitlines()[-1])
            print(w, x["value"], x["ms_per_step"])


if __name__ == "__main__":
    main(sys.argv[1])
