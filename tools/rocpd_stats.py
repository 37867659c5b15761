#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace) as a per-kernel stats table:
   python tools/rocpd_stats.py gpurun_out/prof/x_results.db [after_fraction] > profiles/r01_kernel_stats.md
after_fraction (0..1): only dispatches that start in the last part of the trace (e.g. 0.97: the developed flow at the
end of a long run)."""
import sqlite3
import sys


def main(path, after=0.0):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    t0, t1 = list(db.execute("select min(start), max(start) from kernels"))[0]
    cut = t0 + after * (t1 - t0)
    q = "select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels where start >= %d group by %s order by 3 desc" % (name, int(cut), name)
    rows = list(db.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        n = r[0].split("(")[0]
        print("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (n, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
    print("\ntotal kernel time %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
