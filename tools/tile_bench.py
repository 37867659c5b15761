#!/usr/bin/env python3
"""(needs tools/variants/lds_tile_walk_experiment.patch applied: profiles/r05_lds_tile_experiment.md)
The pressure passes on the LDS tile walk (csrc/tile.h, WCSPH_TILE=1) against the gather walk, ONE context (the two paths see the
same sort: the order of the particles inside a cell comes from atomics and differs between contexts):
   python tools/tile_bench.py [config=C3] [marks=25,1200] [mode=faithful]
At every mark: the slot lists checked against the gather lists entry for entry on the device (tile_check); three pressure
iterations through either path from the same state compared bit for bit (vel, adv_rho); kernel-group replays timed (us per launch)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WCSPH_TILE"] = "1"
import numpy as np
from wcsph_amd import dfsph, scene, _lib

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
marks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "25,1200").split(",")]
mode = sys.argv[3] if len(sys.argv) > 3 else "faithful"
n = scene.CONFIGS.get(cfg) or int(cfg)
lib = _lib.load()
pd = dfsph.init_particle(particleDim=n, neighbor_mode=mode); dfsph.reset_param()
ctx = pd._need_ctx()


def t(name, reps=10):
    v = C.c_float(); _lib.check(ctx, lib.wcsph_time_kernel(ctx, name.encode(), reps, C.byref(v))); return round(v.value * 1e3, 1)


def stats():
    out = (C.c_int32 * 8)(); _lib.check(ctx, lib.wcsph_tile_stats(ctx, C.byref(out))); return list(out)


def tile(on):
    _lib.check(ctx, lib.wcsph_tile_enable(ctx, on))


done = 0
for m in marks:
    tile(0)
    while done < m:
        dfsph.step(); done += 1
    row = {"config": cfg, "step": m}
    pd.hash_grid.update_grid(); dfsph.compute_density(); dfsph.compute_dfsph_coff()
    _lib.check(ctx, lib.wcsph_dfsph_begin_pressure(ctx))
    keep = {"vel": pd.vel.to_numpy(), "adv_rho": pd.adv_rho.to_numpy(), "kappa": dfsph.kappa.to_numpy()}
    res = {}
    for on in (0, 1):
        pd.vel.from_numpy(keep["vel"]); pd.adv_rho.from_numpy(keep["adv_rho"]); dfsph.kappa.from_numpy(keep["kappa"])
        tile(on)
        t("pressure_iter", 1)      # prime + warm-up + 1 = three iterations, the last two through this path
        res[on] = (pd.vel.to_numpy(), pd.adv_rho.to_numpy())
    (va, aa), (vb, ab) = res[0], res[1]
    row["vel_max_abs_diff"] = float(np.abs(va - vb).max()); row["vel_bit_equal"] = bool((va.view(np.uint32) == vb.view(np.uint32)).all())
    row["adv_rho_max_abs_diff"] = float(np.abs(aa - ab).max()); row["adv_rho_bit_equal"] = bool((aa.view(np.uint32) == ab.view(np.uint32)).all())
    st = stats()
    row.update(groups=st[0], table_words=st[1], groups_without_box=st[2])
    t("tile_check", 1)
    st2 = stats()
    row.update(list_mismatch_owners=st2[3] - st[3], list_entries_checked=st2[4] - st[4])
    tile(0)
    for g in ("pressure_iter", "pressure_vel_loop", "pressure_drho_loop", "build_lists"): row[g] = t(g)
    tile(1)
    for g in ("pressure_iter", "tile_vel", "tile_drho", "tile_build"): row["tile_" + g if not g.startswith("tile") else g] = t(g)
    print(json.dumps(row), flush=True)
    # leave the state as a gather-walk run would have it
    pd.vel.from_numpy(keep["vel"]); pd.adv_rho.from_numpy(keep["adv_rho"]); dfsph.kappa.from_numpy(keep["kappa"])
pd.close()
