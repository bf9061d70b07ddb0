#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "pyr" 2>&1 | tail -2
for wv in 0 4 8 16; do echo "== waves $wv"; MSPL_PREP_WAVES=$wv timeout -k 10 200 python tools/bench_ops.py prep 2>&1 | grep prep; done
for d in 1 3; do echo -n "pass depth=$d: "; timeout -k 10 120 python bench.py --profile-pass --in-flight $d --steps 90 --warmup 15 2>&1 | grep -o '"value": [0-9.]*'; done
