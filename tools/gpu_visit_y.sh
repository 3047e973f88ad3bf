#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BARGS="--config 1 --steps 20 --warmup 10 --repeats 1 --no-cpu-baseline --exchange-steps 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o pmc -- python $R/bench.py $BARGS > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o pmc -- python $R/bench.py $BARGS > $O/pmc_write.log 2>&1
python $R/tools/rocpd_summary.py $O $O/summary.txt 1 > /dev/null 2>&1
grep -n "FETCH_SIZE\|WRITE_SIZE" $O/summary.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
cd $R; python bench.py --config 1 --no-cpu-baseline --exchange-steps 0 2>/dev/null | tail -1 | cut -c1-200
