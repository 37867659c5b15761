#!/usr/bin/env python3
"""The pressure loop's working set over a C3 run: stream blocks per particle, the fraction of 128-byte stream lines of
each block row that are touched, pressure_iter replay time.  python tools/ws_over_time.py 25,60,120,300,600,1200"""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wcsph_amd import dfsph, _lib
marks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "25,60,120,300,600,1200").split(",")]
pd = dfsph.init_particle(particleDim=126, neighbor_mode="faithful"); dfsph.reset_param()
lib, ctx = _lib.load(), pd._need_ctx()
done = 0
for m in marks:
    while done < m: dfsph.step(); done += 1
    pd.hash_grid.update_grid(); dfsph.compute_density()
    h = np.zeros((5, 64), np.int64)
    _lib.check(ctx, lib.wcsph_list_stats(ctx, h.ctypes.data))
    b = np.arange(64); npart = h[0].sum(); nwave = h[2].sum()
    blocks = (h[0] * b).sum() / npart
    wave_blocks = (h[2] * b).sum() / nwave          # 1 KB per (wave, block) touched
    v = C.c_float(); _lib.check(ctx, lib.wcsph_time_kernel(ctx, b"pressure_iter", 10, C.byref(v)))
    stream_mb = wave_blocks * 1024 * nwave / 1e6
    print(json.dumps({"step": done, "dt": round(float(dfsph.deltaT[0]), 5), "blocks_per_particle": round(float(blocks), 3), "blocks_per_wave": round(float(wave_blocks), 3),
                      "stream_MB_touched": round(float(stream_mb), 1), "working_set_MB": round(float(stream_mb + 104 * npart / 2000376), 1),
                      "frac_5plus": round(float(h[0][5:].sum() / npart), 4), "pressure_iter_us": round(v.value * 1e3, 1)}), flush=True)
pd.close()
