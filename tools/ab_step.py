#!/usr/bin/env python3
"""In-situ step time of the C3 dam break at a few points of its evolution, nothing else (fast A/B of library builds):
   WCSPH_LIB=build_variants/x.so python tools/ab_step.py [marks, e.g. 25,100,1200] [steps timed per mark]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wcsph_amd import dfsph, scene  # noqa: E402

marks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "25,100,1200").split(",")]
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pd = dfsph.init_particle(particleDim=scene.CONFIGS["C3"], neighbor_mode="faithful")
dfsph.reset_param()
done, out = 0, []
for m in marks:
    while done < m:
        dfsph.step(); done += 1
    pd.sync()
    t0 = time.perf_counter()
    for _ in range(nt):
        dfsph.step(); done += 1
    pd.sync()
    out.append("%d: %.3f" % (m, (time.perf_counter() - t0) / nt * 1e3))
print(os.path.basename(os.environ.get("WCSPH_LIB", "product")), " ms/step at ", "  ".join(out), flush=True)
pd.close()
