#!/bin/bash
# Profile the bench workload on the GPU box: kernel-trace stats + separate PMC passes (rocprofv3), outputs under gpurun_out/$1
TAG=${1:-prof}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 60 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
# the kernel-trace pass runs the SAME command as the bench line above (minus the CPU leg) so the timed launches can be compared 1:1
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --exchange-steps 0 > $OUT/stats.log 2>&1
BARGS="--steps 20 --warmup 10 --no-cpu-baseline --exchange-steps 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- python $R/bench.py $BARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- python $R/bench.py $BARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc_sq -o pmc -- python $R/bench.py $BARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq2 -o pmc -- python $R/bench.py $BARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --kernel-trace -d $OUT/pmc_ic -o pmc -- python $R/bench.py $BARGS > $OUT/pmc_ic.log 2>&1
find $OUT -name "*.csv" | head -30
# trim: keep only stats + counter csvs (drop big traces)
find $OUT -name "*kernel_trace.csv" -size +2M -delete
du -sh $OUT
