#!/usr/bin/env python3
"""Surface reconstruction timing on the C3 dam break: python tools/mc_time.py [config] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wcsph_amd import dfsph, scene
n = scene.CONFIGS.get(sys.argv[1] if len(sys.argv) > 1 else "C3")
pd = dfsph.init_particle(particleDim=n, neighbor_mode="faithful")
dfsph.reset_param()
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    dfsph.step()
pd.hash_grid.update_grid(); dfsph.compute_density(); pd.sync()
mc = pd.mc_grid
mc.cal_surface_point(); pd.sync()
t0 = time.perf_counter(); mc.cal_surface_point(); pd.sync(); t1 = time.perf_counter()
mc.marching_cube(); t2 = time.perf_counter()
print("nodes %d (%s)  surface_value %.2f ms  marching_cube+download %.2f ms  vertices %d" % (mc.grid_num, mc.blocknp[0], (t1 - t0) * 1e3, (t2 - t1) * 1e3, mc.vertex_count[0]))
pd.close()
