#!/usr/bin/env python3
"""Mean PMC counter values of the kernels whose name contains a substring, over several rocpd databases:
   python tools/rocpd_pmc_kernel.py k_vel_from_k gpurun_out/pmc/p*/pmc_results.db"""
import sqlite3
import sys

pat = sys.argv[1]
for path in sys.argv[2:]:
    db = sqlite3.connect(path)
    try:
        rows = list(db.execute("select counter_name, count(*), avg(value), avg(duration) from counters_collection "
                               "where kernel_name like ? group by 1", ("%" + pat + "%",)))
    except Exception as e:
        print(path, "ERR", e)
        continue
    for r in rows:
        print("%-36s n=%-4d avg=%-16.1f dur_us=%.1f" % (r[0], r[1], r[2], (r[3] or 0) / 1e3))
