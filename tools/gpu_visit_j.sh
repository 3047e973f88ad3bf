#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_j
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_trainer.py tests/test_trainer.py -m gpu -q -x > $O/pytest_trainer.log 2>&1; grep -E "passed|failed|Error" $O/pytest_trainer.log | tail -5
echo "== fused"; python tools/trainer_rate.py --iters 1000 --only hip 2>&1 | grep -i "train" | tail -3
echo "== layer-by-layer"; DTRL_TRAINER_FUSED=0 python tools/trainer_rate.py --iters 1000 --only hip 2>&1 | grep -i "train" | tail -3
