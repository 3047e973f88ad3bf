#!/usr/bin/env python3
"""Static check of a hipcc -S dump: scratch (spill) instructions per loop nest of one kernel.
Usage: tools/isa_spills.py file.s kernel_substring"""
import re, sys, collections
src, key = sys.argv[1], sys.argv[2]
inside = False; cur = ("-", 0); agg = collections.Counter(); tot = collections.Counter()
for line in open(src):
    if re.match(r"^[A-Za-z_][\w.$]*:", line) and not line.startswith(".L"):
        inside = key in line
        cur = ("-", 0)
    if not inside: continue
    m = re.search(r"in Loop: Header=(\S+) Depth=(\d+)", line)
    if line.startswith(".LBB") or line.startswith("; %bb"):
        cur = (m.group(1), int(m.group(2))) if m else ("-", 0)
    t = line.strip().split(" ")[0] if line.strip() else ""
    if t.startswith("scratch_"): agg[cur] += 1
    if re.match(r"^(v_|s_|ds_|global_|scratch_|flat_|buffer_)", t): tot[cur] += 1
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print("loop %-12s depth %d: %6d instr, %4d scratch" % (k[0], k[1], v, agg.get(k, 0)))
