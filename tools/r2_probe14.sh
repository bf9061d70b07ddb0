#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py -x -q -k "conv3x3 or conv_fn or conv_affine" 2>&1 | tail -4
for cfg in "MSPL_GC3S=1" "MSPL_GC3S=0"; do
for d in 1 3; do echo -n "$cfg pass depth=$d: "; env $cfg timeout -k 10 120 python bench.py --profile-pass --in-flight $d --steps 90 --warmup 18 2>&1 | grep -o '"value": [0-9.]*'; done
done
