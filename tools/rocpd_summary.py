#!/usr/bin/env python3
"""Summarise a tools/gpu_profile.sh output directory (rocprofv3 rocpd sqlite files) into a small text file for profiles/.
Usage: tools/rocpd_summary.py gpurun_out/<tag> profiles/<name>.txt"""
import json
import os
import sqlite3
import sys

KERNEL = "dtrl_frame_kernel"


def main(src, dst, cfg="1"):
    out = []
    traffic = {}
    bj = os.path.join(src, "bench.json")
    if os.path.exists(bj):
        for line in open(bj):
            line = line.strip()
            if line.startswith("{"):
                out.append("## bench.py line (python bench.py --config %s --steps 60 --warmup 20)\n" % cfg + line + "\n")
    db = os.path.join(src, "stats", "stats_results.db")
    if os.path.exists(db):
        cur = sqlite3.connect(db).cursor()
        out.append("## rocprofv3 --kernel-trace --stats -- python bench.py --config %s --steps 60 --warmup 20 --repeats 3 --no-cpu-baseline --exchange-steps 0" % cfg)
        out.append("%-60s %8s %14s %14s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            out.append("%-60s %8d %14.1f %14.1f %8.3f" % (r[0][:60], r[1], r[2], r[3], r[4]))
        rows = list(cur.execute("select grid_x, count(*), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels where name like '%" + KERNEL + "%' group by grid_x"))
        out.append("\n%s dispatches by grid size (threads): grid, n, avg_ms, min_ms, max_ms, vgpr, sgpr, lds_bytes, scratch_bytes" % KERNEL)
        for r in rows:
            out.append("  %8d %4d %10.3f %10.3f %10.3f %5s %5s %7s %7s" % (r[0], r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5], r[6], r[7], r[8]))
        dom = list(cur.execute("select grid_x from kernels where name like '%" + KERNEL + "%' group by grid_x order by sum(duration) desc limit 1"))
        if dom:
            per_step = 2   # env groups per frame (the engine splits a batch of >= 1024 envs in two)
            last = [r[0] for r in cur.execute("select duration from kernels where name like '%" + KERNEL + "%' and grid_x = ? order by start desc limit ?", (dom[0][0], 180 * per_step))]
            out.append("  timed region = last %d frame launches (grid %d, %d env-group launches per bench step, 3 windows x 60 frames): avg %.3f ms (compare roofline.kernel_avg_ms of the bench line)" % (len(last), dom[0][0], per_step, sum(last) / len(last) / 1e6))
        out.append("")
    vals = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_ic", "pmc_lanes", "pmc_pipes"):
        db = os.path.join(src, sub, "pmc_results.db")
        if not os.path.exists(db):
            continue
        cur = sqlite3.connect(db).cursor()
        out.append("## rocprofv3 --pmc (%s pass) -- per-dispatch averages for %s (frame launches of the env groups only; bench.py --config %s --steps 20 --warmup 10 --repeats 1: pre-roll + warm-up + window)" % (sub, KERNEL, cfg))
        q = ("select counter_name, count(*), avg(value), max(grid_size) from counters_collection where kernel_name like '%" + KERNEL + "%' "
             "and grid_size = (select grid_size from counters_collection where kernel_name like '%" + KERNEL + "%' group by grid_size order by count(*) desc limit 1) group by counter_name order by counter_name")
        for r in cur.execute(q):
            extra = ""
            if r[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                extra = "  (KB; = %.1f MB per launch)" % (r[2] / 1024.0)
            out.append("  %-24s n=%3d avg=%.6g%s" % (r[0], r[1], r[2], extra))
            if r[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                traffic[r[0]] = r[2] * 1024.0
            if r[0] == "SQ_INSTS_VALU":
                traffic[r[0]] = r[2]
            vals[(sub, r[0])] = r[2]
        out.append("")
    # derived figures
    der = []
    lanes = None
    if ("pmc_lanes", "SQ_THREAD_CYCLES_VALU") in vals and vals.get(("pmc_lanes", "SQ_ACTIVE_INST_VALU")):
        lanes = vals[("pmc_lanes", "SQ_THREAD_CYCLES_VALU")] / vals[("pmc_lanes", "SQ_ACTIVE_INST_VALU")]
        der.append("  active lanes per VALU instruction = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = %.1f of 64 (%.0f %%)" % (lanes, 100.0 * lanes / 64.0))
    if ("pmc_sq2", "GRBM_GUI_ACTIVE") in vals:
        try:
            cur = sqlite3.connect(os.path.join(src, "pmc_sq2", "pmc_results.db")).cursor()
            r = list(cur.execute("select avg(duration), count(*) from kernels where name like '%" + KERNEL + "%' and grid_x = (select grid_x from kernels where name like '%" + KERNEL + "%' group by grid_x order by sum(duration) desc limit 1)"))[0]
            ghz = vals[("pmc_sq2", "GRBM_GUI_ACTIVE")] / 8.0 / r[0]     # GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles; duration in ns
            der.append("  shader clock while the kernel ran = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration of the same pass (%.3f ms over %d launches) = %.2f GHz" % (r[0] / 1e6, r[1], ghz))
            json.dump({"shader_clock_ghz": ghz, "active_lanes_per_valu_instruction": lanes, "source": os.path.basename(dst)}, open(os.path.join(os.path.dirname(dst), "shader_clock_config%s.json" % cfg), "w"))
        except Exception as e:   # (older rocpd schemas: no kernels view in a counter pass)
            der.append("  shader clock: not derived (%s)" % e)
    if der:
        out.append("## derived\n" + "\n".join(der) + "\n")
    open(dst, "w").write("\n".join(out) + "\n")
    if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
        # MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B -> doubled; WRITE_SIZE is uncalibrated (taken as is)
        rec = {"source": os.path.basename(dst), "fetch_size_bytes_per_launch_raw": traffic["FETCH_SIZE"], "write_size_bytes_per_launch_raw": traffic["WRITE_SIZE"],
               "hbm_bytes_per_launch": 2.0 * traffic["FETCH_SIZE"] + traffic["WRITE_SIZE"],
               "correction": "2 x FETCH_SIZE + WRITE_SIZE (gfx950 read-request correction of MI355X_MICROARCH.md; separate --pmc passes)",
               "sq_insts_valu_per_launch": traffic.get("SQ_INSTS_VALU"),
               "workload": "python bench.py --config %s --steps 20 --warmup 10 --repeats 1 (env-group frame launches only)" % cfg}
        json.dump(rec, open(os.path.join(os.path.dirname(dst), "hbm_traffic_config%s.json" % cfg), "w"), indent=1)
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "1")
