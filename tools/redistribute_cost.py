#!/usr/bin/env python3
"""Local (torch-side) cost of SlabSim.redistribute for an interior rank: classification of a 2 M-row state tensor, message
assembly and the in-place ownership change that run every step at world > 1 (the exchange itself is not timed here; the
header exchange is represented by one host wait).  `old`: the round-2 form (four compactions, cumsum, rows[stay]).
   python tools/redistribute_cost.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wcsph_amd.slab import SlabSim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000376
dev = torch.device("cuda", 0)
W = 15
base = torch.rand((n + 40000, W), device=dev)
base[:, 0] = base[:, 0] * 60.0            # a slab 60 cell columns wide, cuts at 2 / 58: ~3 % near each cut, ~0.3 % leaving per side
lo, hi, G = 2, 58, 2


def old(rows):
    cx = torch.trunc(rows[:, 0]).to(torch.int64)
    go_l, go_r = cx < 0.2, cx >= 59.8
    stay = ~(go_l | go_r)
    near_l, near_r = stay & (cx < lo + G), stay & (cx >= hi - G)
    lost = (cx < lo - G - 100) | (cx >= hi + G + 100)
    n_ml, n_mr, n_lost = torch.stack([go_l.sum(), go_r.sum(), lost.sum()]).tolist()
    new_index = torch.cumsum(stay.to(torch.int64), 0) - 1
    idx_nl, idx_nr = torch.nonzero(near_l).flatten(), torch.nonzero(near_r).flatten()
    mig_l = rows.index_select(0, torch.nonzero(go_l).flatten())
    mig_r = rows.index_select(0, torch.nonzero(go_r).flatten())
    to_l = torch.cat([mig_l, rows.index_select(0, idx_nl)], 0)
    to_r = torch.cat([mig_r, rows.index_select(0, idx_nr)], 0)
    torch.tensor([to_l.shape[0], n_ml], device=dev).tolist()        # header exchange: a host wait
    owned = torch.cat([rows[stay], to_l[:n_mr], to_r[:n_ml]], 0)      # as many arrive as leave
    s_l = torch.cat([new_index[idx_nl], torch.arange(n_mr, device=dev)]).to(torch.int32)
    s_r = torch.cat([new_index[idx_nr], torch.arange(n_ml, device=dev)]).to(torch.int32)
    return owned, s_l, s_r


def new(rows):
    cx = torch.trunc(rows[:, 0]).to(torch.int64)
    code = torch.zeros(rows.shape[0], dtype=torch.uint8, device=dev)
    code = torch.where(cx < lo + G, 1, code); code = torch.where(cx < 0.2, 3, code); code = torch.where(cx < lo - G - 100, 5, code)
    code = torch.where(cx >= hi - G, 2, code); code = torch.where(cx >= 59.8, 4, code); code = torch.where(cx >= hi + G + 100, 5, code)
    idx = torch.nonzero(code).flatten()
    sub = code[idx].to(torch.int64)
    c = torch.bincount(sub, minlength=6).tolist()
    idx = idx[torch.argsort(sub, stable=True)]
    n_nl, n_nr, n_ml, n_mr = c[1], c[2], c[3], c[4]
    idx_nl, idx_nr, idx_gl, idx_gr = idx[:n_nl], idx[n_nl:n_nl + n_nr], idx[n_nl + n_nr:n_nl + n_nr + n_ml], idx[n_nl + n_nr + n_ml:]
    mig_l, mig_r = rows.index_select(0, idx_gl), rows.index_select(0, idx_gr)
    to_l = torch.cat([mig_l, rows.index_select(0, idx_nl)], 0)
    to_r = torch.cat([mig_r, rows.index_select(0, idx_nr)], 0)
    torch.tensor([to_l.shape[0], n_ml], device=dev).tolist()        # header exchange: a host wait
    arrivals = torch.cat([to_l[:n_mr + 37], to_r[:n_ml]], 0)         # a few more arrive than leave: the append path
    rows2, pos, a, b = SlabSim._swap_rows(None, rows, torch.cat([idx_gl, idx_gr]), arrivals, idx_nl, idx_nr)
    s_l = torch.cat([a, pos[:n_mr + 37]]).to(torch.int32)
    s_r = torch.cat([b, pos[n_mr + 37:]]).to(torch.int32)
    return rows2, s_l, s_r


for name, fn in (("round-2 form", old), ("in-place form", new)):
    for _ in range(5):
        fn(base[:n])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R = 20
    for _ in range(R):
        fn(base[:n])
    torch.cuda.synchronize()
    print("redistribute, local torch work on %d rows, %-14s %7.1f us per step" % (n, name + ":", (time.perf_counter() - t0) / R * 1e6))


# the library's classifier in place of the torch classification (one launch + one host wait)
from wcsph_amd.slab import HipBackend
import numpy as np
B = HipBackend(capacity=1024, solid_local=np.ones((4, 3)), bounds=(np.zeros(3, np.float32), np.full(3, 3.0, np.float32)), window=(0, 61),
               particleR=0.025, device=dev)
base[:, 0] = base[:, 0] / 20.0          # cell columns of 2 R = 0.05


def native(rows):
    c, (idx_nl, idx_nr, idx_gl, idx_gr) = B.classify_rows(rows, lo, hi, G, 1, 1)
    n_ml, n_mr = c[2], c[3]
    mig_l, mig_r = rows.index_select(0, idx_gl), rows.index_select(0, idx_gr)
    to_l = torch.cat([mig_l, rows.index_select(0, idx_nl)], 0)
    to_r = torch.cat([mig_r, rows.index_select(0, idx_nr)], 0)
    torch.tensor([to_l.shape[0], n_ml], device=dev).tolist()
    arrivals = torch.cat([to_l[:n_mr], to_r[:n_ml]], 0)
    rows2, pos, a, b = SlabSim._swap_rows(None, rows, torch.cat([idx_gl, idx_gr]), arrivals, idx_nl, idx_nr)
    return rows2, torch.cat([a, pos[:n_mr]]).to(torch.int32), torch.cat([b, pos[n_mr:]]).to(torch.int32)


with torch.cuda.stream(B.stream):
    for _ in range(5):
        native(base[:n])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        native(base[:n])
    torch.cuda.synchronize()
print("redistribute, local work on %d rows, %-14s %7.1f us per step" % (n, "native classifier:", (time.perf_counter() - t0) / 20 * 1e6))
