#!/usr/bin/env python3
"""Idle gaps between consecutive dispatches of a rocprofv3 kernel trace, grouped by the kernel that precedes the gap:
   python tools/rocpd_gaps.py x_results.db [first_step] [n_steps] [min_gap_us]"""
import collections
import sqlite3
import sys


def main(path, first=25, n=20, min_gap=2.0):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = list(db.execute("select %s, start, end from kernels order by start" % name))
    marks = [r[1] for r in rows if r[0].startswith("k_cell_assign")]
    marks = marks[-(len([1 for r in rows if r[0].startswith("k_build_lists")])):]
    a, b = marks[first], marks[first + n]
    sel = [r for r in rows if a <= r[1] < b]
    g = collections.defaultdict(lambda: [0, 0.0])
    small = 0.0
    for p, q in zip(sel, sel[1:]):
        gap = (q[1] - p[2]) / 1e3
        if gap >= min_gap:
            k = p[0].split("(")[0] + " -> " + q[0].split("(")[0]
            g[k][0] += 1; g[k][1] += gap
        elif gap > 0:
            small += gap
    print("gaps >= %.1f us over %d steps (per step):" % (min_gap, n))
    for k, v in sorted(g.items(), key=lambda kv: -kv[1][1]):
        print("  %-70s %5.2f x  %6.1f us each  %6.1f us/step" % (k, v[0] / n, v[1] / v[0], v[1] / n))
    print("  smaller gaps: %.1f us/step" % (small / n))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]) if len(a) > 2 else 25, int(a[3]) if len(a) > 3 else 20, float(a[4]) if len(a) > 4 else 2.0)
