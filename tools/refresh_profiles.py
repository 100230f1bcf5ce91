#!/usr/bin/env python3
"""Turn the outputs of tools/profile_round.sh <tag> (gpurun_out/<tag>/) into the committed summaries:
profiles/<rNN>_bench.json, <rNN>_kernel_stats.md, <rNN>_pmc.md (HBM and SQ tables; hand-written notes below the marker
"## Notes" are kept), <rNN>_pmc_traffic.json.   usage: python tools/refresh_profiles.py <tag> [rNN]   (rNN defaults to the tag's first 3 chars)"""
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


def main(tag, rnd=None):
    rnd = rnd or tag[:3]
    O = os.path.join(ROOT, "gpurun_out", tag)
    P = os.path.join(ROOT, "profiles")
    bench = open(os.path.join(O, "bench.json")).read().strip().splitlines()[-1]
    d = json.loads(bench)
    open(os.path.join(P, f"{rnd}_bench.json"), "w").write(bench + "\n")
    ks = capture(rocpd_stats.main, os.path.join(O, "trace", "t_results.db")).replace(ROOT + "/", "")
    att = [(m.group(1), int(m.group(2)), float(m.group(3))) for m in re.finditer(r"`(lg_attention_kernel<0, false(?:, (?:true|false))+>|lg_attention_dma_kernel)` \| (\d+) \| [\d.]+ \| ([\d.]+)", ks)]
    att_avg = sum(c * a for _, c, a in att) / max(sum(c for _, c, _ in att), 1)
    head = f"""# Round {int(rnd[1:])} — rocprofv3 --kernel-trace summary (final round-{int(rnd[1:])} kernels)

Command (MI355X box, tools/profile_round.sh {tag}): `rocprofv3 --kernel-trace --stats -d gpurun_out/{tag}/trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-pool --sustained-steps 0` (1 warm-up + 3 fully instrumented + 3 timed steps of 33 frames + 32 pairs, plus the steps of the PCIe-inclusive loop; rocpd sqlite summarised by tools/rocpd_stats.py).

Un-profiled bench.py line from the same box just before: profiles/{rnd}_bench.json ({d['value']} frames/s, {d['ms_per_step']} ms/step; lg_attention avg {d['roofline']['avg_launch_ms'] * 1e3:.1f} us by HIP events in the timed region, {att_avg:.1f} us in this trace over its self (rotary, register-staged) and cross (LDS-DMA) kernels: {', '.join(f'{n} {a:.1f} us x {c}' for n, c, a in att)}).

"""
    open(os.path.join(P, f"{rnd}_kernel_stats.md"), "w").write(head + ks)
    hbm_tab, hj = split(capture(rocpd_pmc.main, [os.path.join(O, "pmc_fetch", "f_results.db"), os.path.join(O, "pmc_write", "w_results.db")]))
    sq_tab, _ = split(capture(rocpd_pmc.main, [os.path.join(O, "pmc_sq", "s_results.db")]))
    # bench.py's roofline.traffic looks the dominant stage up by kernel-name substring: give the attention one merged entry
    am = [v for k, v in hj.items() if k.startswith("lg_attention_kernel<0, false") or k == "lg_attention_dma_kernel"]
    if am:
        tot = sum(v["calls"] for v in am)
        hj["lg_attention_kernel (self + cross variants, call-weighted)"] = {k: sum(v[k] * v["calls"] for v in am) / tot for k in ("FETCH_SIZE", "WRITE_SIZE", "traffic_bytes", "avg_us")} | {"calls": tot}
    out = {"note": "per-launch HBM traffic from rocprofv3 PMC, (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes, bench.py workload (33 frames / 32 pairs "
                   f"per step), final round-{int(rnd[1:])} kernels; source gpurun_out/{tag} (tools/profile_round.sh)",
           "kernels": {k: {"fetch_kib": v.get("FETCH_SIZE"), "write_kib": v.get("WRITE_SIZE"), "traffic_bytes": v.get("traffic_bytes"),
                           "avg_us": v.get("avg_us"), "calls": v.get("calls")} for k, v in hj.items()}}
    json.dump(out, open(os.path.join(P, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
    p = os.path.join(P, f"{rnd}_pmc.md")
    notes = ""
    if os.path.exists(p) and "## Notes" in open(p).read():
        notes = open(p).read()[open(p).read().index("## Notes"):]
    headp = f"""# Round {int(rnd[1:])} — PMC passes (final round-{int(rnd[1:])} kernels)

Separate rocprofv3 runs, kernel-trace only (tools/profile_round.sh {tag}): `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE`; each `-- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-pool --sustained-steps 0`.
FETCH_SIZE / WRITE_SIZE: KiB per dispatch averaged per kernel. HBM traffic per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 B (gfx950 FETCH_SIZE counts half of wide coalesced reads, MI355X guide HBM section; WRITE_SIZE uncalibrated).
SQ rows are per XCD/SE slice: MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (32 * GRBM_GUI_ACTIVE); clock = GRBM_GUI_ACTIVE / duration.

"""
    open(p, "w").write(headp + "## HBM\n" + hbm_tab + "\n\n## SQ / GRBM\n" + sq_tab + "\n\n" + notes)
    print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d.get("pcie_inclusive", {}).get("value"),
          d.get("cpu_baseline", {}).get("value"))
    for w in ("c2", "c3", "c5"):
        f = os.path.join(O, f"lat_{w}.json")
        if os.path.exists(f) and open(f).read().strip():
            x = json.loads(open(f).read().strip().splitlines()[-1])
            print(w, x["value"], x["ms_per_step"])


if __name__ == "__main__":
    main(*sys.argv[1:3])
