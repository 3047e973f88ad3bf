#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/tr -o tr -- python $R/tools/trainer_rate.py --iters 200 --repeats 1 --only hip > $O/tr.log 2>&1
DB=$(find $O/tr -name "*.db" | head -1)
python $R/tools/rocpd_top.py $DB 30 > $O/trainer_top.txt 2>&1
python $R/tools/rocpd_gaps.py $DB > $O/trainer_gaps.txt 2>&1
python - <<PY > $O/trainer_seq.txt 2>&1
import sqlite3
cur = sqlite3.connect("$DB").cursor()
rows = [(s, e, n) for n, s, e in cur.execute("select name, start, end from kernels order by start") if "dtrl_tr::" in n]
# one iteration in the middle: print 80 consecutive kernels with start offsets
k0 = len(rows) // 2
t0 = rows[k0][0]
for s, e, n in rows[k0:k0 + 90]:
    print("%9.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:110]))
PY
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
cat $O/trainer_top.txt | cut -c1-200 | head -30
