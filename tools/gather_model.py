#!/usr/bin/env python3
"""Offline model of what a 64-lane neighbour gather touches: from a dumped state (tools/dump_state.py) rebuild the
cell-sorted order and the in-support lists of a sample of waves on the CPU and count, per list slot, the distinct 64-byte
segments the wave's 64 record addresses fall into (TCP_TOTAL_CACHE_ACCESSES counts one tag look-up per such segment) -- for
the product's list order and for alternatives (entry order inside a list, slot alignment, tile shape).

    python tools/gather_model.py gpurun_out/state_1200.npz [n_waves]
"""
import sys
import numpy as np
from scipy.spatial import cKDTree

f = np.load(sys.argv[1])
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
pos = f["pos"].astype(np.float32)
minb = np.asarray(f["minb"], np.float32).reshape(-1)[:3]
bs = [int(x) for x in f["bs"]]
inv = np.float32(1.0 / 0.05)
H = 0.1
c = np.trunc((pos - minb) * inv).astype(np.int64)
inbox = np.all((c >= 0) & (c < np.array(bs)), 1)
pos, c = pos[inbox], c[inbox]
nx, ny, nz = bs
print("particles in box", len(pos), "grid", bs)


def order_for(T0B, T1B, T2B):
    perm = 1 if ny > nx else 0
    a0 = c[:, 1] if perm else c[:, 0]
    a1 = c[:, 0] if perm else c[:, 1]
    nt1 = ((nx if perm else ny) + (1 << T1B) - 1) >> T1B
    nt2 = (nz + (1 << T2B) - 1) >> T2B
    TS = T0B + T1B + T2B
    lin = (((((a0 >> T0B) * nt1 + (a1 >> T1B)) * nt2) + (c[:, 2] >> T2B)) << TS) | ((a0 & ((1 << T0B) - 1)) << (T1B + T2B)) | ((a1 & ((1 << T1B) - 1)) << T2B) | (c[:, 2] & ((1 << T2B) - 1))
    return np.argsort(lin, kind="stable"), perm


def model(shape=(1, 2, 4), list_order="walk", align=None, seed=1, subcell=None):
    od, perm = order_for(*shape)
    if subcell == "morton":   # deterministic order inside a cell: by the Morton code of the in-cell offset
        fr = ((pos - minb) * inv - c).clip(0, 0.999)
        q = (fr * 8).astype(np.int64)
        m = np.zeros(len(pos), np.int64)
        for b in range(3):
            m |= ((q[:, 0] >> b) & 1) << (3 * b + 2) | ((q[:, 1] >> b) & 1) << (3 * b + 1) | ((q[:, 2] >> b) & 1) << (3 * b)
        od, perm = order_for(*shape)
        # re-sort with the sub-cell key as tie-break
        T0B, T1B, T2B = shape
        a0 = c[:, 1] if perm else c[:, 0]; a1 = c[:, 0] if perm else c[:, 1]
        nt1 = ((nx if perm else ny) + (1 << T1B) - 1) >> T1B; nt2 = (nz + (1 << T2B) - 1) >> T2B; TS = T0B + T1B + T2B
        lin = (((((a0 >> T0B) * nt1 + (a1 >> T1B)) * nt2) + (c[:, 2] >> T2B)) << TS) | ((a0 & ((1 << T0B) - 1)) << (T1B + T2B)) | ((a1 & ((1 << T1B) - 1)) << T2B) | (c[:, 2] & ((1 << T2B) - 1))
        od = np.lexsort((m, lin))
    P, C = pos[od], c[od]
    tree = cKDTree(P.astype(np.float64))
    rng = np.random.default_rng(seed)
    n = len(P)
    w0 = rng.integers(0, n // 64 - 1, nw) * 64
    segs, trips, slots_used, entries = [], [], [], []
    co = C[:, 1] if perm else C[:, 0]
    cm = C[:, 0] if perm else C[:, 1]
    for w in w0:
        idx = np.arange(w, w + 64)
        nb = tree.query_ball_point(P[idx].astype(np.float64), H * 1.000005)
        lists = []
        for l, i in enumerate(idx):
            j = np.array([x for x in nb[l] if x != i], np.int64)
            if list_order == "walk":
                key = np.lexsort((j, cm[j] - cm[i], co[j] - co[i]))
            elif list_order == "index":
                key = np.argsort(j)
            elif list_order == "mid_first":
                key = np.lexsort((j, co[j] - co[i], cm[j] - cm[i]))
            elif list_order == "zrel":       # by z offset of the neighbour's cell, then rows
                key = np.lexsort((j, cm[j] - cm[i], co[j] - co[i], C[j, 2] - C[i, 2]))
            j = j[key]
            lists.append(j)
        ln = np.array([len(x) for x in lists])
        entries.append(ln.mean())
        if align is None:
            kmax = ln.max()
            A = np.full((64, kmax), -1, np.int64)
            for l, x in enumerate(lists):
                A[l, :len(x)] = x
        else:
            # slot alignment: the entries of one group (align = "plane": outer offset; "row": (outer, middle) offset) start at
            # a slot common to all lanes of the wave (wave-level padding)
            groups = []
            for l, x in enumerate(lists):
                i = idx[l]
                if align == "plane16":
                    g = co[x] - co[i] + 2
                elif isinstance(align, tuple):     # a partition of the five outer planes: align[d + 2] = group of plane offset d
                    g = np.array(align)[co[x] - co[i] + 2]
                else:
                    g = (co[x] - co[i] + 2) * (5 if align == "row" else 1) + ((cm[x] - cm[i] + 2) if align == "row" else 0)
                groups.append(g)
            ng = (max(align) + 1) if isinstance(align, tuple) else (25 if align == "row" else 5)
            if align == "plane16":
                # alignment inside each 16-lane brick row only (a gather's coherence is a matter of the lanes that address one
                # source row): a group pads to ITS maximum per plane, the wave runs as many slots as its longest group
                parts = []
                for q in range(4):
                    cols = []
                    for gid in range(5):
                        cnt = [int((groups[l] == gid).sum()) for l in range(16 * q, 16 * q + 16)]
                        m = max(cnt)
                        B = np.full((16, m), -1, np.int64)
                        for l in range(16):
                            x = lists[16 * q + l]
                            sel = x[groups[16 * q + l] == gid]
                            B[l, :len(sel)] = sel
                        cols.append(B)
                    parts.append(np.concatenate(cols, 1))
                kmax = max(p_.shape[1] for p_ in parts)
                A = np.full((64, kmax), -1, np.int64)
                for q, p_ in enumerate(parts):
                    A[16 * q:16 * q + 16, :p_.shape[1]] = p_
            else:
                cols = []
                for gid in range(ng):
                    cnt = [int((g == gid).sum()) for g in groups]
                    m = max(cnt)
                    B = np.full((64, m), -1, np.int64)
                    for l, x in enumerate(lists):
                        sel = x[groups[l] == gid]
                        B[l, :len(sel)] = sel
                    cols.append(B)
                A = np.concatenate(cols, 1)
                kmax = A.shape[1]
        t8 = -(-kmax // 8)
        trips.append(t8)
        slots_used.append(kmax)
        for k in range(kmax):
            a = A[:, k]
            a = a[a >= 0]
            segs.append(len(np.unique(a >> 2)))     # 16-byte records: 4 per 64-byte segment
    segs = np.array(segs)
    cost = np.maximum(segs, 16)
    tot_cost = cost.sum() / nw
    return dict(shape=shape, order=list_order, align=align, subcell=subcell, entries=round(float(np.mean(entries)), 2), slots=round(float(np.mean(slots_used)), 2),
                trips8=round(float(np.mean(trips)), 2), segs_per_gather=round(float(segs.mean()), 2), cost_per_gather=round(float(cost.mean()), 2),
                cycles_per_wave=round(float(tot_cost), 1))


if __name__ == "__main__":
    import json
    base = None
    if len(sys.argv) > 3 and sys.argv[3] == "rest":
        runs = [dict(), dict(align="plane"), dict(shape=(0, 3, 3)), dict(shape=(1, 2, 3)), dict(shape=(2, 2, 2)), dict(shape=(1, 3, 2))]
    elif len(sys.argv) > 3 and sys.argv[3] == "plane16":
        runs = [dict(), dict(align="plane16"), dict(align="plane")]
    elif len(sys.argv) > 3 and sys.argv[3] == "groups":
        runs = [dict(), dict(align=(0, 0, 1, 2, 2)), dict(align=(0, 0, 0, 1, 1)), dict(align=(0, 0, 1, 1, 1)), dict(shape=(1, 3, 2)), dict(shape=(1, 2, 3), align="plane")]
    else:
      runs = [dict(), dict(list_order="index"), dict(list_order="mid_first"), dict(list_order="zrel"), dict(align="plane"), dict(align="row"),
            dict(subcell="morton"), dict(shape=(0, 1, 5)), dict(shape=(0, 0, 6)), dict(shape=(1, 1, 5)), dict(shape=(0, 2, 4)), dict(shape=(0, 3, 3)), dict(shape=(1, 2, 3)), dict(shape=(2, 2, 2))]
    for kw in runs:
        r = model(**kw)
        if base is None:
            base = r["cycles_per_wave"]
        r["rel"] = round(r["cycles_per_wave"] / base, 3)
        print(json.dumps(r), flush=True)
