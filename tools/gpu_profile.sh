#!/bin/bash
# Profile the bench workload on the GPU box: kernel-trace stats + separate PMC passes (rocprofv3), outputs under gpurun_out/$1
#   tools/gpu_profile.sh <tag> [config]     config = 1 (dog slopes_mixed 4096, default) | 2 (raptor narrow_gaps 8192)
TAG=${1:-prof}
CFG=${2:-1}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --config $CFG --steps 60 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json
# the kernel-trace pass runs the SAME command as the bench line above (minus the CPU leg) so the timed launches can be compared 1:1: 3 windows of 60 frames
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --config $CFG --steps 60 --warmup 20 --repeats 3 --no-cpu-baseline --no-trained-leg --no-fp32-leg --exchange-steps 0 > $OUT/stats.log 2>&1
# counter passes: one window of 20 frames behind the same pre-roll (every frame launch of the run is averaged: pre-roll + warm-up + window)
BARGS="--config $CFG --steps 20 --warmup 10 --repeats 1 --no-cpu-baseline --no-trained-leg --no-fp32-leg --exchange-steps 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- python $R/bench.py $BARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- python $R/bench.py $BARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc_sq -o pmc -- python $R/bench.py $BARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq2 -o pmc -- python $R/bench.py $BARGS > $OUT/pmc_sq2.log 2>&1
# lane occupancy of the vector instructions ("23 of 64 lanes" was a model figure): active lanes per VALU instruction = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc_lanes -o pmc -- python $R/bench.py $BARGS > $OUT/pmc_lanes.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA --kernel-trace -d $OUT/pmc_pipes -o pmc -- python $R/bench.py $BARGS > $OUT/pmc_pipes.log 2>&1
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $OUT/sq_counters_available.txt
# the text summary is made on the box (the sqlite files are too large to bring back)
python $R/tools/rocpd_summary.py $OUT $OUT/summary.txt $CFG > /dev/null 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +1M -delete
du -sh $OUT
