#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from rocpd sqlite outputs (one or more passes).
usage: tools/rocpd_pmc.py pass1.db [pass2.db ...]   -> markdown table + JSON (stdout)
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads, so
traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (WRITE_SIZE uncalibrated per the guide)."""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "").replace("rfe::", "")
    return name[:90]


def main(paths):
    agg = {}
    for p in paths:
        db = sqlite3.connect(p)
        for name, cname, val, dur in db.execute("select name, counter_name, counter_value, duration from pmc_events"):
            a = agg.setdefault(short(name), {}).setdefault(cname, [0, 0.0, 0.0])
            a[0] += 1; a[1] += val; a[2] += dur
    counters = sorted({c for k in agg.values() for c in k})
    print("| kernel | calls | avg us | " + " | ".join(counters) + " |")
    print("|---|---:|---:|" + "---:|" * len(counters))
    out = {}
    for k, cs in sorted(agg.items(), key=lambda kv: -max(v[2] for v in kv[1].values())):
        any_c = next(iter(cs.values()))
        row = {c: cs[c][1] / cs[c][0] for c in cs}
        out[k] = dict(calls=any_c[0], avg_us=any_c[2] / any_c[0] / 1e3, **row)
        if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
            out[k]["traffic_bytes"] = (2 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024
        print(f"| `{k}` | {any_c[0]} | {any_c[2] / any_c[0] / 1e3:.1f} | " + " | ".join(f"{row.get(c, float('nan')):.4g}" for c in counters) + " |")
    print("\n```json\n" + json.dumps(out, indent=1) + "\n```")


if __name__ == "__main__":
    main(sys.argv[1:])
