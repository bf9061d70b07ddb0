#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in "1 1" "3 2" "3 1"; do set -- $d; echo -n "depth=$1 group=$2: "; timeout -k 10 120 python bench.py --profile-pass --in-flight $1 --group $2 --steps 90 --warmup 18 2>&1 | grep -o '"value": [0-9.]*'; done
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
