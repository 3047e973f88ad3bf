R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v11_rates; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python tools/config_rates.py > $O/config_rates.txt 2>&1; cat $O/config_rates.txt
timeout 400 python tools/scale_n.py 4096 16384 65536 > $O/scale_host.txt 2>&1; cat $O/scale_host.txt
TERRAIN_GEN=device timeout 400 python tools/scale_n.py 4096 16384 65536 > $O/scale_dev.txt 2>&1; cat $O/scale_dev.txt
timeout 600 python tools/soak.py 4096 2000 > $O/soak.txt 2>&1; tail -2 $O/soak.txt
timeout 300 python bench.py --no-cpu-baseline --exchange-steps 0 --steps 60 --warmup 20 --terrain-gen device > $O/bench_dev.log 2>&1; python - <<'PY'
import json
for l in open('/root/repo/gpurun_out/v11_rates/bench_dev.log'):
    if l.startswith('{'):
        d=json.loads(l); print("device-terrain bench:", d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])
PY
