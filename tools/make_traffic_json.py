#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a tools/pmc_traffic.sh summary of pressure_iter (FETCH_SIZE / WRITE_SIZE in KB per
dispatch, separate --pmc runs): bytes per launch = 2 x FETCH (gfx950 correction, MI355X_MICROARCH "HBM") + WRITE, summed
over the two kernels of the pair; stamped with the kernel source hash bench.py checks.
   python tools/make_traffic_json.py gpurun_out/r02c_pmc_traffic.txt profiles/r02c_pmc_pressure_iter.md"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
txt = open(sys.argv[1]).read()
cur, vals = None, {}
for line in txt.splitlines():
    m = re.match(r"## (\w+)", line)
    if m: cur = m.group(1); continue
    m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\d+\s+avg=([\d.]+)", line)
    if m: vals.setdefault(m.group(1), []).append(float(m.group(2)))
f, w = sum(vals["FETCH_SIZE"]), sum(vals["WRITE_SIZE"])
b = int((2 * f + w) * 1024)
out = {"n_liquid": 2000376, "pressure_iter_bytes_per_launch": b, "kernel_source_sha16": bench.kernel_source_sha16(),
       "source": sys.argv[2] if len(sys.argv) > 2 else sys.argv[1],
       "note": "FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE x1, KB per dispatch; separate --pmc passes; fused solver-loop kernels"}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"))
print(out)
