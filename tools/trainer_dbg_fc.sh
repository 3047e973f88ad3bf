cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06fc; mkdir -p $O
for dbg in 2 3 0; do
  DTRL_TRAINER_DBG=$dbg DTRL_TRAINER_FUSED=4 rocprofv3 --kernel-trace --stats -d $O/s$dbg -o stats -- python $R/tools/trainer_rate.py --iters 100 --repeats 1 --only hip > $O/s$dbg.log 2>&1
  echo "== stage $dbg"; python $R/tools/rocpd_top.py $(find $O/s$dbg -name "*.db" | head -1) 12 | grep "fused_backward"
  find $O/s$dbg -name "*.db" -delete
done
