#!/usr/bin/env python3
"""Launch shape of the kernels of a rocprofv3 kernel trace whose name contains a pattern: python tools/rocpd_kernel_shape.py db pattern"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print(cols)
name = "name" if "name" in cols else "kernel_name"
pat = sys.argv[2]
seen = {}
for r in db.execute("select * from kernels where %s like '%%%s%%' limit 4000" % (name, pat)):
    d = dict(zip(cols, r))
    key = tuple((k, d[k]) for k in cols if any(t in k.lower() for t in ("grid", "workgroup", "lds", "scratch", "vgpr", "sgpr", "queue", "stream")) )
    seen[key] = seen.get(key, 0) + 1
for k, v in seen.items():
    print(v, dict(k))
