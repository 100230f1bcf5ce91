#!/usr/bin/env python3
"""Inter-kernel idle time in a rocprofv3 (rocpd sqlite) kernel trace: over the last `frac` of the dispatches, the sum of
kernel durations, the sum of the gaps between consecutive kernels (gaps above `cap` us, i.e. step boundaries / host
syncs, are reported separately) and the gap histogram.  Used to decide whether launch-bound paths need a hipGraph.
usage: tools/rocpd_gaps.py trace_results.db [frac=0.5] [cap_us=50]"""
import sqlite3
import sys


def main(path, frac=0.5, cap=50.0):
    db = sqlite3.connect(path)
    rows = sorted(db.execute("select start, end from kernels").fetchall())
    rows = rows[int(len(rows) * (1 - frac)):]
    busy = sum(e - s for s, e in rows) / 1e3
    gaps = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1)]
    small = [g for g in gaps if g <= cap]
    big = [g for g in gaps if g > cap]
    print(f"{len(rows)} dispatches: kernel time {busy / 1e3:.3f} ms, gaps <= {cap} us: {sum(small) / 1e3:.3f} ms "
          f"(mean {sum(small) / max(len(small), 1):.2f} us, n={len(small)}), larger gaps: {sum(big) / 1e3:.3f} ms (n={len(big)})")
    for lo, hi in ((-1e9, 0), (0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, cap)):
        print(f"  gap ({lo if lo > -1e8 else '-inf'}, {hi}] us: {sum(1 for g in small if lo < g <= hi)}")


if __name__ == "__main__":
    main(sys.argv[1], *(float(a) for a in sys.argv[2:]))
