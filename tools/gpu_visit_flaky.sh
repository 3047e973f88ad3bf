#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do python -m pytest tests -m gpu -q -x 2>&1 | grep -a "passed\|failed" | tail -1; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-260
