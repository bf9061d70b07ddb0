#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "eesp or dw" 2>&1 | tail -2
timeout -k 10 200 python tools/bench_ops.py k2 2>&1 | tail -5
for d in 1 3; do echo -n "pass depth=$d: "; timeout -k 10 120 python bench.py --profile-pass --in-flight $d --steps 90 --warmup 15 2>&1 | grep -o '"value": [0-9.]*'; done
