#!/bin/bash
# One GPU-box visit, parametrised (replaces the lettered one-off scripts of rounds 2-4):  gpurun -- 'bash tools/gpu_visit.sh <tag> <stage> [<stage> ...]'
# Stages: tests [-k expr via $K] | smoke | bench | bench2 (configs[2]) | prof1 | prof2 (rocprofv3 kernel trace + PMC passes of configs[1] / [2]) | sections |
#         trainer (rate + kernel table) | loops (training loops end to end) | learn_dog | learn_raptor (tools/learn_curve.py) | ab:<VAR>=<a>,<b> (same-box bench A/B)
# Everything lands under gpurun_out/<tag>/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for stage in "$@"; do
  case $stage in
    tests) python -m pytest tests -m gpu -q -x ${K:+-k "$K"} > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log ;;
    tests_all) python -m pytest tests -m gpu -q ${K:+-k "$K"} > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log ;;
    bench) python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; cat $O/bench.json ;;
    bench_fast) python bench.py --no-cpu-baseline 2> $O/bench_fast.err | tail -1 > $O/bench_fast.json; cat $O/bench_fast.json ;;
    bench2) python bench.py --config 2 --no-cpu-baseline 2> $O/bench2.err | tail -1 > $O/bench2.json; cat $O/bench2.json ;;
    prof1) bash tools/gpu_profile.sh ${TAG}_cfg1 1 > $O/prof1.log 2>&1; tail -2 $O/prof1.log ;;
    prof2) bash tools/gpu_profile.sh ${TAG}_cfg2 2 > $O/prof2.log 2>&1; tail -2 $O/prof2.log ;;
    sections) python tools/gpu_sections.py 4096 60 1 > $O/sections_cfg1.txt 2>&1; python tools/gpu_sections.py 8192 60 2 > $O/sections_cfg2.txt 2>&1; tail -3 $O/sections_cfg1.txt ;;
    trainer) python tools/trainer_rate.py --iters 1000 > $O/trainer_rate.log 2>&1; grep Train $O/trainer_rate.log ;;
    loops)
      for spec in "args/opt_args_train_mace.txt 4096" "args/opt_args_train_raptor_mace.txt 8192"; do
        set -- $spec
        echo "== $1 envs=$2 frames=600 trainer=hip --overlap" >> $O/train_loops.log
        python tools/train_mace.py --arg-file $1 --envs $2 --frames 600 --trainer hip --overlap 2>&1 | tail -2 >> $O/train_loops.log
      done; cat $O/train_loops.log ;;
    learn_dog) python tools/learn_curve.py --char dog --out $O/learning_curve_dog.txt ${LEARN_ARGS} > $O/learn_dog.log 2>&1; tail -5 $O/learn_dog.log ;;
    learn_goat) python tools/learn_curve.py --char goat --out $O/learning_curve_goat.txt ${LEARN_ARGS} > $O/learn_goat.log 2>&1; tail -5 $O/learn_goat.log ;;
    learn_raptor) python tools/learn_curve.py --char raptor --out $O/learning_curve_raptor.txt ${LEARN_ARGS} > $O/learn_raptor.log 2>&1; tail -5 $O/learn_raptor.log ;;
    ab:*)
      spec=${stage#ab:}; var=${spec%%=*}; vals=${spec#*=}
      for v in ${vals//,/ }; do for rep in 1 2; do echo -n "$var=$v rep $rep: " >> $O/ab_$var.txt; env $var=$v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O/ab_$var.txt; done; done; cat $O/ab_$var.txt ;;
    model_ab)
      for margs in "" "warm_start=0" "contact_breaking=0" "warm_start=0,contact_breaking=0"; do for cfg in 1 2; do
        echo -n "config $cfg model-args [$margs]: " >> $O/model_ab.txt
        python bench.py --config $cfg --no-cpu-baseline --no-trained-leg --no-fp32-leg --exchange-steps 0 --no-rccl-leg ${margs:+--model-args $margs} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['timed_window']['resets_per_frame'])" >> $O/model_ab.txt
      done; done; cat $O/model_ab.txt ;;
    *) echo "unknown stage $stage" ;;
  esac
done
