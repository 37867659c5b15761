#!/usr/bin/env python3
"""Between two slab steps of a tools/ring_real.py trace (RING_NOSYNC=1, one plane): from the end of a step's k_end_press_update_pos to
the first k_cell_assign of the next step's grid update -- the migration + ghost rebuild and whatever the host needs in between.
Span, GPU-busy time inside it, and the kernels that ran there (count and microseconds per boundary).
   python tools/rocpd_step_boundary.py /tmp/prof/ring_results.db"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = [(n.split("(")[0], s, e) for n, s, e in db.execute("select %s, start, end from kernels order by start" % name)]
ends = [k for k, r in enumerate(rows) if r[0] == "k_end_press_update_pos"]
bounds = []
for k in ends:
    nxt = next((q for q in range(k + 1, len(rows)) if rows[q][0] == "k_cell_assign"), None)
    if nxt is not None and not any(rows[q][0] == "k_end_press_update_pos" for q in range(k + 1, nxt)):
        bounds.append((k, nxt))
bounds = bounds[len(bounds) // 2:]
agg = defaultdict(lambda: [0, 0.0])
span = busy = 0.0
for k, nxt in bounds:
    t0, t1 = rows[k][2], rows[nxt][1]
    span += t1 - t0
    ce = t0
    for q in range(k + 1, nxt):
        n, s, e = rows[q]
        agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
        busy += max(0, e - max(s, ce)); ce = max(ce, e)
n = max(len(bounds), 1)
print("%d step boundaries (end of k_end_press_update_pos -> next k_cell_assign): span %.1f us, GPU busy inside %.1f us, idle %.1f us; %d launches per boundary"
      % (len(bounds), span / n / 1e3, busy / n / 1e3, (span - busy) / n / 1e3, round(sum(v[0] for v in agg.values()) / n)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  %7.1f us  %5.2f x  %s" % (v[1] / n, v[0] / n, k))
