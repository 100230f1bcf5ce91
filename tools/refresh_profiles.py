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
import pmc_wait_table
import rocpd_pmc
import rocpd_stats
import stamp


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
    # identity of the build the run was taken from (tools/stamp.py before the gpurun call, completed on the GPU box by profile_round.sh)
    st = json.load(open(os.path.join(O, "stamp.json"))) if os.path.exists(os.path.join(O, "stamp.json")) else {}
    stamp_line = stamp.line(st) if st else "Build: NOT STAMPED (run tools/stamp.py before tools/profile_round.sh)."
    now = stamp.current()
    if st and (st.get("so_sha256") != now["so_sha256"] or st.get("box_so_sha256") not in (None, st.get("so_sha256"))):
        stamp_line += "  **STALE: the library in the tree is no longer the one this run measured.**"
        print("WARNING:", stamp_line, file=sys.stderr)
    stamp_line += "\n"
    bench = open(os.path.join(O, "bench.json")).read().strip().splitlines()[-1]
    d = json.loads(bench)
    open(os.path.join(P, f"{rnd}_bench.json"), "w").write(bench + "\n")
    ks = capture(rocpd_stats.main, os.path.join(O, "trace", "t_results.db")).replace(ROOT + "/", "")
    att = [(m.group(1), int(m.group(2)), float(m.group(3))) for m in re.finditer(r"`(lg_attention_kernel<0, false(?:, (?:true|false))+>|lg_attention_dma_kernel)` \| (\d+) \| [\d.]+ \| ([\d.]+)", ks)]
    att_avg = sum(c * a for _, c, a in att) / max(sum(c for _, c, _ in att), 1)
    head = f"""# Round {int(rnd[1:])} — rocprofv3 --kernel-trace summary (final round-{int(rnd[1:])} kernels)

Command (MI355X box, tools/profile_round.sh {tag}): `rocprofv3 --kernel-trace --stats -d gpurun_out/{tag}/trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-pool --sustained-steps 0` (1 warm-up + 3 fully instrumented + 3 timed steps of 33 frames + 32 pairs, plus the steps of the PCIe-inclusive loop; rocpd sqlite summarised by tools/rocpd_stats.py).

Un-profiled bench.py line from the same box just before: profiles/{rnd}_bench.json ({d['value']} frames/s, {d['ms_per_step']} ms/step; lg_attention avg {d['roofline']['avg_launch_ms'] * 1e3:.1f} us by HIP events in the timed region, {att_avg:.1f} us in this trace: {', '.join(f'{n} {a:.1f} us x {c}' for n, c, a in att)}).

"""
    open(os.path.join(P, f"{rnd}_kernel_stats.md"), "w").write(head + stamp_line + "\n" + ks)
    hbm_tab, hj = split(capture(rocpd_pmc.main, [os.path.join(O, "pmc_fetch", "f_results.db"), os.path.join(O, "pmc_write", "w_results.db")]))
    sq_tab, _ = split(capture(rocpd_pmc.main, [os.path.join(O, "pmc_sq", "s_results.db")]))
    # bench.py's roofline.traffic looks the dominant stage up by kernel-name substring: give the attention one merged entry
    am = [v for k, v in hj.items() if k.startswith("lg_attention_kernel<0, false") or k == "lg_attention_dma_kernel"]
    if len(am) > 1:   # rounds 2-4: self blocks on the register-staged rotary kernel, cross blocks on the LDS-DMA kernel (round 5: every launch is lg_attention_dma_kernel)
        tot = sum(v["calls"] for v in am)
        hj["lg_attention_kernel (self + cross variants, call-weighted)"] = {k: sum(v[k] * v["calls"] for v in am) / tot for k in ("FETCH_SIZE", "WRITE_SIZE", "traffic_bytes", "avg_us")} | {"calls": tot}
    out = {"note": "per-launch HBM traffic from rocprofv3 PMC, (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes, bench.py workload (33 frames / 32 pairs "
                   f"per step), final round-{int(rnd[1:])} kernels; source gpurun_out/{tag} (tools/profile_round.sh)",
           "build": st,
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
    open(p, "w").write(headp + stamp_line + "\n## HBM\n" + hbm_tab + "\n\n## SQ / GRBM\n" + sq_tab + "\n\n" + notes)
    # ---- latency configurations: un-profiled lines + kernel traces of the same commands
    lat = [f"# Round {int(rnd[1:])} -- latency configurations (BASELINE configs[1] / [2] / [4]): `bench.py --workload c2|c3|c5 --steps 100 --warmup 10` and the kernel "
           f"trace of the same command at 30 steps (tools/profile_round.sh {tag})\n", stamp_line]
    for w in ("c2", "c3", "c5"):
        f, g = os.path.join(O, f"lat_{w}.json"), os.path.join(O, f"stats_{w}.md")
        if os.path.exists(f) and open(f).read().strip():
            x = json.loads(open(f).read().strip().splitlines()[-1])
            lat.append(f"## {w}: {x['ms_per_step']} ms per step ({x['value']} {x['unit']}); with an event pair around every stage {x.get('ms_per_step_with_stage_events')} ms\n")
            lat.append("stage table (ms per step, events around every stage): " + ", ".join(f"{k} {v}" for k, v in x.get("stages_ms_per_step", {}).items()) + "\n")
        if os.path.exists(g):
            lines = open(g).read().replace(ROOT + "/", "").splitlines()
            lat.append("\n".join(lines[:34]) + "\n")
            tl = [ln for ln in lines if ln.startswith("Timeline:")]
            if tl:
                import re as _re
                m = _re.search(r"Timeline: ([0-9.]+) ms", tl[0])
                steps_traced = 65            # 30 steps + 5 warm-up, two passes (plain + with stage events)
                lat.append(tl[0] + f"\n\n(= {float(m.group(1)) / steps_traced:.3f} ms per traced step with at least one kernel running, against the UNTRACED wall time above: the ~10 us gaps are the "
                           "profiler's per-dispatch completion handling, the untraced stream has none of that size -- a kernel boundary costs 2.4-4.3 us all in, "
                           f"profiles/{rnd}_grid_barrier.md.)\n")
    # one pair / one stereo frame with RFE_OPT_LG_FP16X2 on (same binary, same box)
    for w in ("c3", "c5"):
        f = os.path.join(O, f"lat_{w}_fp16x2.json")
        if os.path.exists(f) and open(f).read().strip():
            x = json.loads(open(f).read().strip().splitlines()[-1])
            lat.append(f"## {w} with RFE_OPT_LG_FP16X2 = 1 (default off): {x['ms_per_step']} ms per step ({x['value']} {x['unit']}); with an event pair around every stage "
                       f"{x.get('ms_per_step_with_stage_events')} ms\n")
            lat.append("stage table (ms per step, events around every stage): " + ", ".join(f"{k} {v}" for k, v in x.get("stages_ms_per_step", {}).items()) + "\n")
    g = os.path.join(O, "stats_c3_fp16x2.md")
    if os.path.exists(g):
        lat.append("kernel trace of c3 with the option on (`gemm_lat_kernel<..., true>` / `lg_attention_lat_kernel<4, true>` = the split forms):\n")
        lat.append("\n".join(open(g).read().replace(ROOT + "/", "").splitlines()[:24]) + "\n")
    open(os.path.join(P, f"{rnd}_latency_kernel_stats.md"), "w").write("\n".join(lat))
    # ---- RFE_OPT_LG_FP16X2 diagnostic configuration (same binary, same box, same call)
    fb = os.path.join(O, "bench_fp16x2.json")
    if os.path.exists(fb) and open(fb).read().strip():
        line = open(fb).read().strip().splitlines()[-1]
        open(os.path.join(P, f"{rnd}_bench_fp16x2.json"), "w").write(line + "\n")
        x = json.loads(line)
        body = open(os.path.join(O, "stats_fp16x2.md")).read().replace(ROOT + "/", "") if os.path.exists(os.path.join(O, "stats_fp16x2.md")) else ""
        clk = open(os.path.join(O, "pmc_sq_fp16x2.md")).read() if os.path.exists(os.path.join(O, "pmc_sq_fp16x2.md")) else ""
        clk = clk[:clk.index("```json")] if "```json" in clk else clk

        def mhz(table, want):
            """kernel name prefix -> (avg us, GRBM_GUI_ACTIVE) from a rocpd_pmc table whose header names the columns"""
            cols = None
            for ln in table.splitlines():
                cells = [c.strip() for c in ln.strip().strip("|").split("|")]
                if "GRBM_GUI_ACTIVE" in cells:
                    cols = (cells.index("avg us"), cells.index("GRBM_GUI_ACTIVE"))
                elif cols and cells and cells[0].startswith("`" + want):
                    return float(cells[cols[0]]), float(cells[cols[1]])
            return None
        default_pmc = open(os.path.join(P, f"{rnd}_pmc.md")).read() if os.path.exists(os.path.join(P, f"{rnd}_pmc.md")) else ""
        note = ""
        rows = []
        for k in ("conv1ab_fused_kernel", "conv3x3_mfma_kernel<64, false, true, 2, 8>", "conv3x3_mfma_kernel<128, true, true, 5, 8>"):
            a, b = mhz(default_pmc, k), mhz(clk, k)
            if a and b:
                rows.append(f"| `{k}` | {a[0]:.1f} | {a[1]:.4g} | {a[1] / a[0]:.0f} | {b[0]:.1f} | {b[1]:.4g} | {b[1] / b[0]:.0f} |")
        if rows:
            note = ("\nThe SuperPoint kernels are the SAME code in both configurations; they take the same number of shader cycles and run at a lower clock when the option is on "
                    "(consistent with power management over the whole step: its f16 matrix phases draw more; the clock is not a property of the kernel):\n\n"
                    "| kernel | default: avg us | GRBM_GUI_ACTIVE | MHz | fp16x2 on: avg us | GRBM_GUI_ACTIVE | MHz |\n|---|---:|---:|---:|---:|---:|---:|\n" + "\n".join(rows) + "\n")
        open(os.path.join(P, f"{rnd}_fp16x2_kernel_stats.md"), "w").write(
            f"# Round {int(rnd[1:])} -- rocprofv3 --kernel-trace summary with RFE_OPT_LG_FP16X2 ON (diagnostic run; the default / headline configuration is "
            f"{rnd}_kernel_stats.md)\n\n{stamp_line}\nUn-profiled line of the same configuration from the same box, same call (`bench.py --steps 20 --warmup 3 --lg-fp16x2 1 --no-pool "
            f"--no-pcie`): profiles/{rnd}_bench_fp16x2.json ({x['value']} frames/s, {x['ms_per_step']} ms/step).\n\n" + body +
            "\n## Clocks in this configuration (`--pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES`; clock = GRBM_GUI_ACTIVE / duration; compare the SQ / GRBM "
            f"table of {rnd}_pmc.md for the default configuration)\n\n" + clk + note)
    # ---- what the waves wait on: throughput step, single pair, fp16x2
    tags = [t for t in (tag + "_wait", tag + "_wait_c3") if os.path.exists(os.path.join(ROOT, "gpurun_out", t, "table.md"))]
    if len(tags) == 2:
        pmc_wait_table.main(rnd, tags[0], tags[1])
        pw = os.path.join(P, f"{rnd}_pmc_wait.md")
        t = open(pw).read()
        extra = ""
        fx = os.path.join(ROOT, "gpurun_out", tag + "_wait_fp16x2", "table.md")
        if os.path.exists(fx):
            extra = "\n## throughput step with RFE_OPT_LG_FP16X2 = 1\n\n" + open(fx).read()
        open(pw, "w").write(t.replace("\n", "\n" + stamp_line + "\n", 1) + extra)
    print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d.get("pcie_inclusive", {}).get("value"),
          d.get("cpu_baseline", {}).get("value"))
    for w in ("c2", "c3", "c5"):
        f = os.path.join(O, f"lat_{w}.json")
        if os.path.exists(f) and open(f).read().strip():
            x = json.loads(open(f).read().strip().splitlines()[-1])
            print(w, x["value"], x["ms_per_step"])


if __name__ == "__main__":
    main(*sys.argv[1:3])
