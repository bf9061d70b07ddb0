#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "pipelined" 2>&1 | tail -5
for g in 1 2 3; do echo -n "group $g: "; timeout -k 10 160 python bench.py --profile-pass --in-flight 3 --group $g --steps 60 --warmup 12 2>&1 | grep -o '"value": [0-9.]*'; done
echo -n "group 2 odd steps: "; timeout -k 10 160 python bench.py --profile-pass --in-flight 3 --group 2 --steps 51 --warmup 11 2>&1 | grep -o '"value": [0-9.]*'
