#!/usr/bin/env python3
"""GPU busy fraction and the gaps between dispatches of the LAST part of a rocprofv3 kernel trace (steady state):
   python tools/rocpd_busy2.py /tmp/prof/x_results.db [tail_fraction=0.5]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = list(db.execute("select %s, start, end from kernels order by start" % name))
t0, t1 = rows[0][1], rows[-1][1]
seg = [r for r in rows if r[1] >= t0 + (1 - frac) * (t1 - t0)]
steps = sum(1 for r in seg if "k_end_press_update_pos" in r[0])
span = seg[-1][2] - seg[0][1]
busy, cs, ce = 0, seg[0][1], seg[0][2]
gaps = []
for i in range(1, len(seg)):
    n, s, e = seg[i]
    if s > ce:
        gaps.append(((s - ce) / 1e3, seg[i - 1][0].split("(")[0], n.split("(")[0]))
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("steps %d  dispatches %d (%.1f per step)  span %.3f ms (%.1f us per step)  busy %.1f %%  idle per step %.1f us" % (
    steps, len(seg), len(seg) / max(steps, 1), span / 1e6, span / 1e3 / max(steps, 1), 100.0 * busy / span, (span - busy) / 1e3 / max(steps, 1)))
gaps.sort(reverse=True)
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for g, a, b in gaps:
    agg[(a, b)][0] += 1; agg[(a, b)][1] += g
print("idle by kernel boundary (us per step, count per step):")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %7.1f  %5.2f  %s -> %s" % (v[1] / max(steps, 1), v[0] / max(steps, 1), k[0], k[1]))
