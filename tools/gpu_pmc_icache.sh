#!/bin/bash
# instruction-fetch counters of the bench workload (separate rocprofv3 --pmc passes), output under gpurun_out/$1
TAG=${1:-icache}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BARGS="--steps 10 --warmup 5 --no-cpu-baseline"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_ic -o pmc -- python $R/bench.py $BARGS > $OUT/pmc_ic.log 2>&1
rocprofv3 --pmc SQC_TC_INST_REQ SQC_TC_STALL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc_ic2 -o pmc -- python $R/bench.py $BARGS > $OUT/pmc_ic2.log 2>&1
python3 - <<PY
import sqlite3, glob
for sub in ("pmc_ic", "pmc_ic2"):
    for db in glob.glob("$OUT/%s/*.db" % sub):
        cur = sqlite3.connect(db).cursor()
        q = ("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%dtrl_frame_kernel%' "
             "and grid_size = (select grid_size from counters_collection where kernel_name like '%dtrl_frame_kernel%' group by grid_size order by count(*) desc limit 1) group by counter_name order by counter_name")
        for r in cur.execute(q): print("%-30s n=%3d avg=%.6g" % r)
PY
