#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do
python tools/train_mace.py --distributed --data-parallel --overlap --envs 4096 --frames 600 --trainer hip 2>&1 | grep -a "distributed x" | head -1
python tools/train_mace.py --distributed --overlap --envs 4096 --frames 600 --trainer hip 2>&1 | grep -a "distributed x" | head -1
python tools/train_mace.py --overlap --envs 4096 --frames 600 --trainer hip 2>&1 | grep -a "env-steps/s" | tail -1
done
