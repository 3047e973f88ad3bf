#!/usr/bin/env python3
"""Trainer kernels under a rocprofv3 --kernel-trace run (rocpd sqlite): per kernel family the launch count, average duration, and the average GAP between a
trainer kernel's start and the end of the trainer kernel before it (launch-to-launch idle time on the trainer's stream: queueing for a wavefront slot shows up
here), split by whether a frame kernel was resident at the kernel's start.   tools/rocpd_gaps.py <results.db>"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qid = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute("select name, start, end, %s from kernels order by start" % (qid or "0")))
frames = [(s, e) for n, s, e, q in rows if "dtrl_frame_kernel" in n]
tr = [(n, s, e, q) for n, s, e, q in rows if "dtrl_tr::" in n]
print("columns:", cols)
print("frame kernels %d (avg %.3f ms), trainer kernels %d, queues used by the trainer: %s, by the frame kernel: %s" % (
    len(frames), sum(e - s for s, e in frames) / max(len(frames), 1) / 1e6, len(tr), sorted(set(q for *_, q in tr)), sorted(set(q for n, s, e, q in rows if "dtrl_frame_kernel" in n))))
import bisect
fs = sorted(frames)
starts = [s for s, e in fs]
def resident(t):
    i = bisect.bisect_right(starts, t)
    return sum(1 for s, e in fs[max(0, i - 4):i] if s <= t < e)
stat = {}
prev_end = None
for n, s, e, q in tr:
    key = resident(s)
    d = stat.setdefault(key, [0, 0.0, 0.0, 0])
    d[0] += 1; d[1] += e - s
    if prev_end is not None and s - prev_end < 200e3:      # (longer pauses are the host's: between Train() calls)
        d[2] += max(0, s - prev_end); d[3] += 1
    prev_end = e
for k in sorted(stat):
    c, dur, gap, ng = stat[k]
    print("frame kernels resident at start = %d: %6d trainer kernels, avg duration %6.2f us, avg gap to the previous trainer kernel %6.2f us (n=%d)" % (k, c, dur / c / 1e3, gap / max(ng, 1) / 1e3, ng))
