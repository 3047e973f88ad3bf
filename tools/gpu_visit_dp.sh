#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_hip_trainer.py -m gpu -q -x -k "data_parallel" 2>&1 | tail -30
