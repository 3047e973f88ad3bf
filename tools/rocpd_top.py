#!/usr/bin/env python3
"""Top-kernel table of a rocprofv3 --kernel-trace --stats run (rocpd sqlite output): tools/rocpd_top.py <results.db> [n]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
print("%-72s %8s %12s %10s %8s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))   # (the top_kernels view is in microseconds)
for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit ?", (n,)):
    print("%-72s %8d %12.2f %10.2f %8.3f" % (r[0][:72], r[1], r[2] / 1e3, r[3], r[4]))
try:
    print("\nGEMM launches by grid (x, y, z): n, avg_us")
    for r in cur.execute("select grid_x, grid_y, grid_z, count(*), avg(duration) from kernels where name like '%tr_gemm%' group by grid_x, grid_y, grid_z order by sum(duration) desc limit 40"):
        print("  grid %6d %4d %4d   n=%6d  avg %8.2f us" % (r[0], r[1], r[2], r[3], r[4] / 1e3))
except Exception as e:
    print("(no per-grid table: %r)" % e)
