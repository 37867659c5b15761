#!/usr/bin/env python3
"""Inside the slab steps of a tools/overlap_ring.py trace (from the first k_cell_assign / k_density_coff_warm of a step to its
k_end_press_update_pos: the ring's own redistribute between steps is excluded): span, busy time of the compute stream, and the idle
gaps by kernel boundary.   python tools/rocpd_ring_steps.py /tmp/prof/ring_results.db"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = [(n.split("(")[0], s, e) for n, s, e in db.execute("select %s, start, end from kernels order by start" % name)]
isr = lambda n: "nccl" in n.lower() or "rccl" in n.lower()
comp = [r for r in rows if not isr(r[0]) and "rocclr_copyBuffer" not in r[0]]
# steps: from the k_density_coff_warm of a step to the k_end_press_update_pos that follows
steps, cur = [], None
for r in comp:
    if r[0] == "k_density_coff_warm": cur = [r]
    elif cur is not None:
        cur.append(r)
        if r[0] == "k_end_press_update_pos": steps.append(cur); cur = None
steps = steps[len(steps) // 2:]
agg = defaultdict(lambda: [0, 0.0])
tot_span = tot_busy = 0.0
for st in steps:
    span = st[-1][2] - st[0][1]
    busy, cs, ce = 0, st[0][1], st[0][2]
    for i in range(1, len(st)):
        n, s, e = st[i]
        if s > ce:
            agg[(st[i - 1][0], n)][0] += 1; agg[(st[i - 1][0], n)][1] += (s - ce) / 1e3
            busy += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    busy += ce - cs
    tot_span += span; tot_busy += busy
n = max(len(steps), 1)
print("%d steps (density pass -> end of step): span %.1f us per step, compute stream busy %.1f us, idle %.1f us" % (len(steps), tot_span / n / 1e3, tot_busy / n / 1e3, (tot_span - tot_busy) / n / 1e3))
print("idle by kernel boundary (us per step, count per step):")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %7.1f  %5.2f  %s -> %s" % (v[1] / n, v[0] / n, k[0], k[1]))
