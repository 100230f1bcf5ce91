#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace as a --stats style table:
per kernel name: calls, total / average / min / max duration, share of GPU kernel time.
usage: tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.md"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("rfe::", "")
    return name[:110]


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {ncol}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"source: {path}  ({len(rows)} dispatches, {tot / 1e6:.3f} ms total kernel time)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{short(name)}` | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | {a[3] / 1e3:.1f} | {100.0 * a[1] / tot:.1f} |")


    if "--gaps" in sys.argv:
        gaps(rows)


def gaps(rows):
    """How much of a step is NOT kernel time: idle intervals between consecutive dispatches (union over streams).  Gaps above 100 us are the
    host side between steps (synchronise, timing, next call) and are listed apart."""
    iv = sorted((s, e) for _, s, e in rows)
    busy = 0; cur_s, cur_e = iv[0]; small = []; big = []
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            (small if s - cur_e < 100_000 else big).append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    small.sort()
    n = len(small)
    print(f"\nTimeline: {busy / 1e6:.3f} ms with at least one kernel running; {n} idle gaps < 100 us between consecutive dispatches, total {sum(small) / 1e6:.3f} ms "
          f"(median {small[n // 2] / 1e3 if n else 0:.2f} us, 90th percentile {small[int(n * 0.9)] / 1e3 if n else 0:.2f} us, max {small[-1] / 1e3 if n else 0:.2f} us) "
          f"= {100.0 * sum(small) / max(busy + sum(small), 1):.1f} % of the in-step time; {len(big)} gaps >= 100 us (between steps / calls), total {sum(big) / 1e6:.3f} ms.")


if __name__ == "__main__":
    main(sys.argv[1])
