#!/usr/bin/env python3
"""Does the halo exchange of the native data plane run BESIDE the pass on the slab interior?  
Reads a rocprofv3 --kernel-trace database of tools/overlap_ring.py (WCSPH_RING_TRACE=1) and, for every RCCL kernel (the grouped
ncclSend / ncclRecv of one halo refresh, the all-reduces), finds what the compute stream did meanwhile:

   python tools/rocpd_overlap.py /tmp/prof/ring_results.db [skip_fraction=0.4] > profiles/r06_overlap.md

A halo refresh inside wcsph_slab_step is   k_halo_pack -> [pass on the interior] -> k_halo_unpack -> [pass on the rest];
its RCCL kernel starts behind the pack (event) and the unpack waits for it (event).  Per refresh:
   hidden   = the part of the RCCL kernel's interval during which a compute kernel was running
   exposed  = the rest (the compute stream idle, waiting for the rows)
and whether the RCCL interval lies inside [start of the interior pass - launch slack, end of the interior pass]."""
import sqlite3, sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = [(n.split("(")[0], s, e) for n, s, e in db.execute("select %s, start, end from kernels order by start" % name)]
t0, t1 = rows[0][1], rows[-1][1]
rows = [r for r in rows if r[1] >= t0 + skip * (t1 - t0)]


def is_rccl(n):
    n = n.lower()
    return "nccl" in n or "rccl" in n


comp = [r for r in rows if not is_rccl(r[0])]
rc = [r for r in rows if is_rccl(r[0])]
steps = sum(1 for r in comp if "k_end_press_update_pos" in r[0])
print("# halo / compute overlap of the native data plane (loop-back ring, one GPU)\n")
print("trace: %d compute dispatches, %d RCCL kernels, %d steps (first %.0f %% of the trace skipped)\n" % (len(comp), len(rc), steps, skip * 100))

# compute intervals, sorted by start; for an RCCL interval: the overlapped time and the kernel that overlaps most
import bisect
starts = [r[1] for r in comp]
kinds = defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0, 0])   # by (rccl kernel class, companion): n, dur, hidden, exposed, inside, preceded by pack
rows_out = []
for n, s, e in rc:
    i = bisect.bisect_left(starts, s)
    j = max(i - 8, 0)
    hid, best, bestn, bs, be = 0, 0, "-", None, None
    segs = []
    while j < len(comp) and comp[j][1] < e:
        cs, ce = comp[j][1], comp[j][2]
        o = min(e, ce) - max(s, cs)
        if o > 0:
            segs.append((max(s, cs), min(e, ce)))
            if o > best:
                best, bestn, bs, be = o, comp[j][0], cs, ce
        j += 1
    segs.sort()
    cur_s, cur_e = None, None
    for a, b in segs:
        if cur_e is None or a > cur_e:
            if cur_e is not None: hid += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    if cur_e is not None: hid += cur_e - cur_s
    dur = e - s
    # the compute kernel that ended last before this RCCL kernel started (a halo refresh: k_halo_pack)
    k = bisect.bisect_right(starts, s) - 1
    prev = "-"
    pe = 0
    for q in range(max(k - 6, 0), k + 1):
        if comp[q][2] <= s + 2000 and comp[q][2] > pe:
            pe, prev = comp[q][2], comp[q][0]
    halo = prev.startswith("k_halo_pack")
    inside = bs is not None and e <= be + 1000
    kind = ("halo refresh" if halo else "other (all-reduce / not behind a pack)", bestn if halo else "-")
    a_ = kinds[kind]
    a_[0] += 1; a_[1] += dur; a_[2] += hid; a_[3] += dur - hid; a_[4] += 1 if inside else 0
    rows_out.append((n, dur, hid, bestn, inside, halo))

print("| RCCL kernels | companion on the compute stream (largest overlap) | count / step | avg us | hidden us | exposed us | ends inside the companion |")
print("|---|---|---|---|---|---|---|")
tot_h = tot_x = 0.0
for (kind, compn), v in sorted(kinds.items(), key=lambda kv: -kv[1][1]):
    print("| %s | %s | %.2f | %.1f | %.1f | %.1f | %d of %d |" % (kind, compn, v[0] / max(steps, 1), v[1] / v[0] / 1e3, v[2] / v[0] / 1e3, v[3] / v[0] / 1e3, v[4], v[0]))
    if kind.startswith("halo"):
        tot_h += v[2]; tot_x += v[3]
print("\nhalo refreshes per step: hidden behind compute %.1f us, exposed %.1f us (%.0f %% of the RCCL time hidden)" % (
    tot_h / 1e3 / max(steps, 1), tot_x / 1e3 / max(steps, 1), 100.0 * tot_h / max(tot_h + tot_x, 1)))
names = defaultdict(int)
for n, *_ in rows_out: names[n] += 1
print("\nRCCL kernel names: " + ", ".join("%s x %d" % kv for kv in names.items()))
# GPU busy over the window (any stream)
allr = sorted(rows, key=lambda r: r[1])
busy, cs, ce = 0, allr[0][1], allr[0][2]
for n, s, e in allr[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
span = allr[-1][2] - allr[0][1]
print("\nwindow: %.3f ms per step, GPU busy %.1f %%" % (span / 1e6 / max(steps, 1), 100.0 * busy / span))
