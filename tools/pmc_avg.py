#!/usr/bin/env python3
"""Per-dispatch averages of the counters of a rocprofv3 --pmc pass (rocpd sqlite) for the frame kernel's full-size launches.
Usage: pmc_avg.py <dir containing *_results.db> [label]"""
import glob, os, sqlite3, sys
KERNEL = "dtrl_frame_kernel"
for db in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)):
    cur = sqlite3.connect(db).cursor()
    q = ("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%" + KERNEL + "%' "
         "and grid_size = (select grid_size from counters_collection where kernel_name like '%" + KERNEL + "%' group by grid_size order by count(*) desc limit 1) group by counter_name order by counter_name")
    for name, cnt, avg in cur.execute(q):
        extra = "  (KB -> %.1f MB per launch)" % (avg / 1024.0) if name in ("FETCH_SIZE", "WRITE_SIZE") else ""
        print("%s %-16s n=%3d avg=%.6g%s" % (sys.argv[2] if len(sys.argv) > 2 else "", name, cnt, avg, extra))
