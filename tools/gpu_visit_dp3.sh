#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_hip_trainer.py -m gpu -q -x 2>&1 | grep -a "passed\|failed" | tail -1
for i in 1 2; do
python tools/train_mace.py --distributed --data-parallel --overlap --envs 4096 --frames 600 --trainer hip 2>&1 | grep -a "distributed x" | head -1
done
