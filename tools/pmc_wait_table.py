#!/usr/bin/env python3
"""Turn the two PMC passes of tools/pmc_wait.sh into the "what the waves wait on" table (profiles/rNN_pmc_wait.md).
usage: python tools/pmc_wait_table.py <rNN> <tag of the throughput run> [<tag of the single-pair (--workload c3) run>]
       python tools/pmc_wait_table.py --table <tag>     (GPU box, end of tools/pmc_wait.sh: writes gpurun_out/<tag>/table.md, which is what
                                                         travels back -- the sqlite files of a latency run exceed the 64 MiB pull limit)"""
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rocpd_pmc import short


def load(tag):
    agg = {}
    for sub, db in (("w1", "a_results.db"), ("w2", "b_results.db")):
        path = None
        for r, _, files in os.walk(os.path.join(ROOT, "gpurun_out", tag, sub)):
            if db in files:
                path = os.path.join(r, db)
        if not path:
            raise SystemExit(f"gpurun_out/{tag}/{sub}/**/{db} not found (run tools/pmc_wait.sh {tag} on the GPU box)")
        con = sqlite3.connect(path)
        for name, cname, val, dur in con.execute("select name, counter_name, counter_value, duration from pmc_events"):
            a = agg.setdefault(short(name), {}).setdefault((sub, cname), [0, 0.0, 0.0])
            a[0] += 1; a[1] += val; a[2] += dur
    return agg


def table(agg, min_share=0.004):
    tot = sum(v[("w1", "SQ_WAVE_CYCLES")][2] for v in agg.values() if ("w1", "SQ_WAVE_CYCLES") in v)
    rows = []
    for k, c in agg.items():
        if ("w1", "SQ_WAVE_CYCLES") not in c or ("w2", "SQ_WAVE_CYCLES") not in c:
            continue
        n, _, dur = c[("w1", "SQ_WAVE_CYCLES")]
        if dur / tot < min_share:
            continue
        g = lambda p, name: c.get((p, name), [1, 0.0, 0.0])[1] / max(c.get((p, name), [1, 0.0, 0.0])[0], 1)
        wc1, wc2 = g("w1", "SQ_WAVE_CYCLES"), g("w2", "SQ_WAVE_CYCLES")
        rows.append((dur, f"| `{k}` | {n} | {dur / n / 1e3:.1f} | {g('w1', 'SQ_WAIT_INST_ANY') / wc1:.2f} | {g('w1', 'SQ_WAIT_ANY') / wc1:.2f} | "
                          f"{g('w1', 'SQ_WAIT_INST_LDS') / wc1:.3f} | {g('w2', 'SQ_ACTIVE_INST_VALU') / wc2:.2f} | {g('w2', 'SQ_ACTIVE_INST_LDS') / wc2:.3f} | "
                          f"{g('w2', 'SQ_ACTIVE_INST_VMEM') / wc2:.3f} | {g('w1', 'SQ_VALU_MFMA_BUSY_CYCLES') / (32 * max(g('w1', 'SQ_BUSY_CYCLES'), 1)):.2f} |"))
    head = ("| kernel | calls | avg µs | wait-to-issue | waitcnt/barrier | LDS port | VALU+MFMA issue | LDS issue | VMEM issue | MFMA busy / (32·SQ busy) |\n"
            "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
    return head + "\n".join(r for _, r in sorted(rows, key=lambda x: -x[0]))


def tag_table(tag):
    f = os.path.join(ROOT, "gpurun_out", tag, "table.md")
    return open(f).read().rstrip() if os.path.exists(f) else table(load(tag))


def main(rnd, tag, tag_c3=None):
    if rnd == "--table":
        open(os.path.join(ROOT, "gpurun_out", tag, "table.md"), "w").write(table(load(tag)) + "\n")
        return
    out = [f"# Round {int(rnd[1:])} — what the waves wait on (rocprofv3 PMC, two passes; tools/pmc_wait.sh {tag}, table by tools/pmc_wait_table.py)\n",
           "Per kernel, sums over the SQ slices per dispatch (averaged): fraction of the resident wave-cycles (SQ_WAVE_CYCLES) spent waiting to issue",
           "an instruction (SQ_WAIT_INST_ANY: for these kernels the matrix pipe being busy), blocked in `s_waitcnt` / `s_barrier` (SQ_WAIT_ANY), waiting on",
           "the LDS issue port (SQ_WAIT_INST_LDS), and issuing VALU (incl. MFMA) / LDS / VMEM instructions (SQ_ACTIVE_INST_*).  Bench workload (33 frames + 32 pairs),",
           "2 profiled steps, kernels of the final binary of the round; kernels below 0.4 % of the profiled time are left out.\n",
           "## throughput step\n", tag_table(tag)]
    if tag_c3:
        out += ["\n## single pair (`--workload c3`)\n", tag_table(tag_c3)]
    open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_wait.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out)[:3000])


if __name__ == "__main__":
    main(*sys.argv[1:4])
