import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
from wcsph_amd import dfsph, _lib
steps = int(sys.argv[1])
pd = dfsph.init_particle(particleDim=126, neighbor_mode="faithful"); dfsph.reset_param()
for _ in range(steps): dfsph.step()
lib, ctx = _lib.load(), pd._need_ctx()
pd.hash_grid.update_grid(); dfsph.compute_density()
out = {}
for g in sys.argv[2:]:
    v = C.c_float(); _lib.check(ctx, lib.wcsph_time_kernel(ctx, g.encode(), 10, C.byref(v))); out[g] = round(v.value * 1e3, 1)
print(steps, out)
