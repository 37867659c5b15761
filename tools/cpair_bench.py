#!/usr/bin/env python3
"""(needs tools/variants/cluster_pair_walk_experiment.patch applied and a -DWCSPH_CPAIR build: tools/build_variant.sh cpair_c2_t4 -DWCSPH_CPAIR -DCP_C=2 -DCP_T=4;
WCSPH_LIB=build_variants/cpair_c2_t4.so CP_C=2)
The density-change pass of pressure_iter on CLUSTER windows (csrc/cpair.h) against the product's per-particle gather walk, ONE
context, at rest and in the developed flow (profiles/r06_cluster_pair.md):
   WCSPH_LIB=build_variants/cpair.so python tools/cpair_bench.py [config=C3] [marks=25,1200] [mode=faithful]
At every mark: both kernels from the same state, adv_rho compared; us per launch of either, of the prototype list builder and of
the product's k_build_lists; windows per cluster and pair efficiency (useful pairs / evaluated pairs)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wcsph_amd import dfsph, scene, _lib

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
marks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "25,1200").split(",")]
mode = sys.argv[3] if len(sys.argv) > 3 else "faithful"
n = scene.CONFIGS.get(cfg) or int(cfg)
lib = _lib.load()
lib.wcsph_cpair_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64 * 8)]
pd = dfsph.init_particle(particleDim=n, neighbor_mode=mode); dfsph.reset_param()
ctx = pd._need_ctx()
nl = pd.liquid_count


def t(name, reps=10):
    v = C.c_float(); _lib.check(ctx, lib.wcsph_time_kernel(ctx, name.encode(), reps, C.byref(v))); return round(v.value * 1e3, 1)


done = 0
for m in marks:
    while done < m:
        dfsph.step(); done += 1
    row = {"config": cfg, "mode": mode, "step": m, "n_liquid": nl}
    pd.hash_grid.update_grid(); dfsph.compute_density(); dfsph.compute_dfsph_coff()
    _lib.check(ctx, lib.wcsph_dfsph_begin_pressure(ctx))
    t("pressure_iter", 1)                        # primes the (pos, k) records; two iterations
    t("drho_press_rec", 1); a = pd.adv_rho.to_numpy().copy()
    row["cpair_build_us"] = t("cpair_build", 1)
    st = (C.c_int64 * 8)(); _lib.check(ctx, lib.wcsph_cpair_stats(ctx, C.byref(st)))
    CPC = int(os.environ.get("CP_C", "4")); nq = (nl + CPC - 1) // CPC
    row.update(windows_per_cluster=round(st[0] / nq, 2), max_windows=int(st[3]), overflowed_clusters=int(st[4]), list_entries_per_particle=round(st[1] / nl, 2),
               duplicate_entries=int(st[2]), pair_efficiency=round(st[1] / max(float(CPC * CPC) * st[0], 1), 3), cluster=CPC,
               gathers_per_particle_cluster=round(2 * st[0] / nq, 1), gathers_per_particle_lists=round(2 * st[1] / nl, 1))
    t("cpair_drho", 1); b = pd.adv_rho.to_numpy().copy()
    d = np.abs(a - b)
    row.update(adv_rho_max_abs_diff=float(d.max()), adv_rho_beyond_1e_5=int((d > 1e-5).sum()), adv_rho_mean=float(a.mean()))
    for g in ("drho_press_rec", "cpair_drho", "pressure_vel_loop", "build_lists"):
        row[g + "_us"] = t(g)
    row["cpair_over_gather"] = round(row["cpair_drho_us"] / row["drho_press_rec_us"], 3)
    print(json.dumps(row), flush=True)
pd.close()
