#!/usr/bin/env python3
"""GPU busy fraction of a rocprofv3 kernel trace (rocpd sqlite): union of kernel intervals / wall span, for the
dispatches after the last `k_alias_fill` (setup) and before the first timing-group replay.
   python tools/rocpd_busy.py /tmp/prof/x_results.db [first_n_steps]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = list(db.execute("select %s, start, end from kernels order by start" % name))
# keep the main loop: from the first k_cell_assign after setup to the last k_update_pos
names = [r[0].split("(")[0] for r in rows]
last_setup = max((i for i, n in enumerate(names) if "k_alias_fill" in n or "k_cell_solid_near" in n), default=0)
ups = [i for i, n in enumerate(names) if n.endswith("k_update_pos")]
lo, hi = last_setup + 1, ups[-1]
seg = rows[lo:hi + 1]
span = seg[-1][2] - seg[0][1]
busy, cur_s, cur_e = 0, seg[0][1], seg[0][2]
for _, s, e in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
gaps = sorted(((seg[i + 1][1] - seg[i][2]) / 1e3, names[lo + i], names[lo + i + 1]) for i in range(len(seg) - 1))
print("steps %d  dispatches %d  span %.3f ms  busy %.3f ms  (%.1f %%)  idle per step %.1f us" % (
    len(ups), len(seg), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e3 / max(len(ups), 1)))
print("largest gaps (us, after kernel -> before kernel):")
for g in gaps[-12:][::-1]:
    print("  %8.1f  %s -> %s" % g)
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for g, a, b in gaps:
    if g > 0: agg[(a, b)][0] += 1; agg[(a, b)][1] += g
print("gap totals by kernel pair (count, total us):")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:15]:
    print("  %5d %9.1f  %s -> %s" % (v[0], v[1], k[0], k[1]))
