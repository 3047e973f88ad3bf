#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for m in 0 1 2; do echo "== DTRL_TRAINER_FUSED=$m"; DTRL_TRAINER_FUSED=$m python tools/trainer_rate.py --iters 1000 --only hip 2>&1 | grep -i "train()" | tail -1; done
timeout 900 python -m pytest tests/test_hip_trainer.py tests/test_trainer.py -m gpu -q -x 2>&1 | tail -2
