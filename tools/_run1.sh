set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lds_fwd; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python tools/gpu_sections.py 4096 30 > $O/sections.log 2>&1; tail -45 $O/sections.log
timeout 300 python bench.py --no-cpu-baseline --exchange-steps 0 --steps 40 --warmup 10 > $O/bench.log 2>&1; tail -c 1500 $O/bench.log
cd /tmp && export TMPDIR=/tmp
for m in net nonet; do for c in WRITE_SIZE FETCH_SIZE; do
 timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${m}_$c -o pmc -- python $R/tools/traffic_probe.py $m > $O/pmc_${m}_$c.log 2>&1
 python $R/tools/pmc_avg.py $O/pmc_${m}_$c $m | tee -a $O/traffic.txt
 find $O/pmc_${m}_$c -name "*.db" -size +8M -delete
done; done
