#!/usr/bin/env python3
"""Per-kernel mean of the PMC counters in a rocprofv3 rocpd database:  python tools/rocpd_pmc.py x_results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
print("# columns:", cols)
name = "kernel_name" if "kernel_name" in cols else "name"
q = "select %s, counter_name, count(*), avg(value), sum(value) from counters_collection group by 1, 2 order by 5 desc" % name
for r in db.execute(q):
    print("%-60s %-14s n=%-5d avg=%.1f" % (r[0].split("(")[0][:60], r[1], r[2], r[3]))
