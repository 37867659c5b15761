#!/usr/bin/env python3
"""Step span vs kernel time from a rocprofv3 kernel trace (rocpd database): steps are delimited by k_build_lists.
   python tools/rocpd_steps.py x_results.db [first_step] [n_steps]  ->  span / kernel sum / idle per step, us"""
import sqlite3
import sys


def main(path, first=25, n=20):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = list(db.execute("select %s, start, end from kernels order by start" % name))
    marks = [r[1] for r in rows if r[0].startswith("k_cell_assign")]
    # one liquid k_cell_assign per step (the first dispatch of a grid update); the static boundary sort runs once, before step 0
    marks = marks[-(len([1 for r in rows if r[0].startswith("k_build_lists")])):]
    a, b = marks[first], marks[first + n]
    ksum = sum(r[2] - r[1] for r in rows if a <= r[1] < b)
    nk = sum(1 for r in rows if a <= r[1] < b)
    print("steps %d..%d: span %.1f us/step, kernel time %.1f us/step, idle %.1f us/step, %.1f dispatches/step"
          % (first, first + n - 1, (b - a) / n / 1e3, ksum / n / 1e3, (b - a - ksum) / n / 1e3, nk / n))


if __name__ == "__main__":
    main(sys.argv[1], *[int(x) for x in sys.argv[2:4]])
