#!/usr/bin/env python3
"""Randomised check of the compact neighbour streams against the CPU oracle (test infrastructure, not collected by pytest: run by hand on a GPU box):
for several scenes / jitters / modes, after every DFSPH step the decoded stream of EVERY liquid particle must equal the
in-support part of the oracle's list (entries within 1e-5 of the list radius may fall on either side).
    python tests/stress_lists.py [seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from util import make_pair

H = 0.1
bad = checked = 0
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for seed in range(seeds):
    for n in (5, 9, 13):
        for jitter in (0.0, 0.45, 0.9):
            for mode in ("faithful", "clean"):
                g, o = make_pair(n, mode, jitter=jitter, seed=1000 + seed)
                rng = np.random.default_rng(seed)
                v = rng.normal(0, 1.5, (g.pd.liquid_count, 3)).astype(np.float32)
                g.pd.vel.from_numpy(v); o.vel[...] = v
                g.call("dfsph_reset_param"); o.reset_param()
                g.pd.vel.from_numpy(v); o.vel[...] = v
                for step in range(4):
                    g.pd.hash_grid.update_grid(); o.update_grid()
                    nl = g.pd.liquid_count
                    assert np.array_equal(g.pd.hash_grid.neighborCount.to_numpy(), o.neighborCount), (n, jitter, mode, step)
                    counts, lists = g.pd.hash_grid.neighbor.compact_rows(0, nl, cap=256)
                    for i in range(nl):
                        mine = np.sort(lists[i][lists[i] >= 0])
                        ref = o.neighbor[i, :o.neighborCount[i]]
                        d = o.pos[ref] - o.pos[i]
                        r2 = np.einsum("ij,ij->i", d, d)
                        want = np.sort(ref[r2 <= np.float32(H * H * 1.00001)])
                        checked += 1
                        if len(mine) != len(want) or not np.array_equal(mine, want):
                            sd = set(mine.tolist()) ^ set(want.tolist())
                            if not all(abs(np.linalg.norm(o.pos[j] - o.pos[i]) - H) < 2e-5 for j in sd):
                                bad += 1
                                print("MISMATCH", n, jitter, mode, step, i, sorted(sd)[:6], flush=True)
                    g.call("dfsph_step"); o.dfsph_step()
                    # keep both sides on the same trajectory
                    g.pd.pos.from_numpy(o.pos.copy()) if False else None
                g.close(); o.close()
                print("ok n=%d jitter=%.2f %s seed=%d  (checked %d lists, %d bad)" % (n, jitter, mode, seed, checked, bad), flush=True)
print("TOTAL checked %d bad %d" % (checked, bad))
sys.exit(1 if bad else 0)
