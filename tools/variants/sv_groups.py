import ctypes as C, os, sys, json
sys.path.insert(0, "/root/repo")
from wcsph_amd import dfsph, _lib
pd = dfsph.init_particle(particleDim=126, neighbor_mode="faithful"); dfsph.reset_param()
lib, ctx = _lib.load(), pd._need_ctx()
for _ in range(45): dfsph.step()
pd.hash_grid.update_grid(); dfsph.compute_density()
row = {}
for g in sys.argv[1:]:
    v = C.c_float(); _lib.check(ctx, lib.wcsph_time_kernel(ctx, g.encode(), 10, C.byref(v))); row[g] = round(v.value * 1e3, 1)
print(json.dumps(row))
