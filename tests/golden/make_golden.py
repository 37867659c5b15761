#!/usr/bin/env python3
"""Generates tests/golden/*.npz: inputs and expected outputs of the hot path on a tiny scene.

The reference (Python + Taichi) cannot be imported or run in the build container and ships no golden
vectors, so these fixtures are produced by the CPU restatement in oracle/ (PARITY UNPINNED by the
reference; pinned only by the analytic known-answers in tests/test_oracle_known_answers.py).
They freeze the oracle's behaviour (regression pin) and give the GPU tests a second, stored target.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import Oracle  # noqa: E402
from wcsph_amd import scene  # noqa: E402


def dfsph_case(mode, n=7, seed=20240901, jitter=0.2):
    liquid, solid = scene.dam_break_dfsph(n)
    rng = np.random.default_rng(seed)
    liquid = liquid + rng.uniform(-jitter, jitter, liquid.shape) * scene.PARTICLE_RADIUS
    pos = np.concatenate([liquid, solid]).astype(np.float32)
    o = Oracle(pos, len(liquid), 0.025, mode=mode)
    o.reset_param()
    out = {"pos0": pos, "n_liquid": np.int64(len(liquid))}
    o.update_grid()
    out["neighborCount0"] = o.neighborCount.copy()
    for s in (1, 2, 3):
        o.dfsph_step()
        for f in ("rho", "alpha_coff", "adv_rho", "kappa", "kappa_v", "vel", "omega", "vel_guess"):
            out["%s_%d" % (f, s)] = getattr(o, f).copy()
        out["pos_%d" % s] = o.pos[:len(liquid)].copy()
        out["scalars_%d" % s] = np.array([o.deltaT, o.vs_iter, o.dv_iter, o.pr_iter], np.float64)
    return out


def sesph_case(mode):
    liquid, solid = scene.sesph_scene(dim=8, boundary=1.0)   # reduced sesph.py:66-92 recipe
    pos = np.concatenate([liquid, solid]).astype(np.float32)
    o = Oracle(pos, len(liquid), 0.05, mode=mode)             # the DIAMETER (Q11)
    o.sesph_init(0.025)
    o.sesph_reset_param()
    out = {"pos0": pos, "n_liquid": np.int64(len(liquid))}
    for s in (1, 2):
        o.sesph_step()
        out["rho_%d" % s] = o.rho.copy(); out["pressure_%d" % s] = o.pressure.copy()
        out["vel_%d" % s] = o.vel.copy(); out["pos_%d" % s] = o.pos[:len(liquid)].copy()
    return out


if __name__ == "__main__":
    for mode in ("faithful", "clean"):
        np.savez_compressed(os.path.join(HERE, "dfsph_7cubed_%s.npz" % mode), **dfsph_case(mode))
        np.savez_compressed(os.path.join(HERE, "sesph_8cubed_%s.npz" % mode), **sesph_case(mode))
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
