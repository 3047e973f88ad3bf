#!/usr/bin/env python3
"""Text timeline of a few outer frames from a rocprofv3 --kernel-trace [--memory-copy-trace] run of the training loop (rocpd sqlite): every dispatch / copy
between two consecutive frame launches of env group 0, times in us relative to the first; trainer kernels folded into runs.   tools/rocpd_timeline.py <db> [first_frame] [n]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
f0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ev = [(s, e, n, q) for n, s, e, q in cur.execute("select name, start, end, queue_id from kernels")]
try:
    ev += [(s, e, "COPY " + str(n), -1) for n, s, e in cur.execute("select name, start, end from memory_copies")]
except Exception as ex:
    print("(no memory copy table: %r)" % ex)
ev.sort()
fk = [x for x in ev if "dtrl_frame_kernel" in x[2]]
q0 = fk[0][3]
starts = [x[0] for x in fk if x[3] == q0]
t0, t1 = starts[f0], starts[f0 + nf]
run = None
def flush():
    global run
    if run:
        print("%9.1f %9.1f  q%-3d  %d trainer kernels (busy %.1f us)" % ((run[0] - t0) / 1e3, (run[1] - t0) / 1e3, run[3], run[2], run[4] / 1e3))
    run = None
for s, e, n, q in ev:
    if s < t0 or s >= t1:
        continue
    if "dtrl_tr::" in n:
        if run and s - run[1] < 30e3:
            run = [run[0], max(run[1], e), run[2] + 1, q, run[4] + (e - s)]
        else:
            flush(); run = [s, e, 1, q, e - s]
        continue
    flush()
    print("%9.1f %9.1f  q%-3d  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, n[:90]))
flush()
