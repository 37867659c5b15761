import ctypes as C, os, sys, json
sys.path.insert(0, "/root/repo")
import numpy as np
from wcsph_amd import dfsph, _lib
lib = _lib.load()
m, n = int(sys.argv[1]), int(sys.argv[2])
def run_to(m):
    pd = dfsph.init_particle(particleDim=n, neighbor_mode="faithful"); dfsph.reset_param()
    for _ in range(m): dfsph.step()
    pd.hash_grid.update_grid(); dfsph.compute_density()
    return pd
out = {}
for path in ("pressure_drho_loop", "pressure_drho_box"):
    pd = run_to(m); ctx = pd._need_ctx()
    v = C.c_float(); _lib.check(ctx, lib.wcsph_time_kernel(ctx, path.encode(), 1, C.byref(v)))
    out[path] = (pd.adv_rho.to_numpy(), pd.pos.to_numpy()[:pd.liquid_count], pd.hash_grid.neighborCount.to_numpy())
    pd.close()
a, pos, nc = out["pressure_drho_loop"]; b = out["pressure_drho_box"][0]
bad = np.where(np.abs(a - b) > 1e-6 * np.abs(a).max())[0]
print("differing", len(bad), "of", len(a))
print("pos of first", pos[bad[:8]], "a", a[bad[:8]], "b", b[bad[:8]], "nc", nc[bad[:8]])
print("pos range all", pos.min(0), pos.max(0))
