#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_supervised.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -12
for cfg in "MSPL_TRAIN_FUSED_FWD=1" "MSPL_TRAIN_FUSED_FWD=0"; do
echo "== $cfg"
env $cfg timeout -k 10 300 python bench.py --no-cpu-baseline --no-three-source --no-io --no-aspp --no-bs64 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('train', d.get('train_step', {}).get('ms_per_step'), 'sup', d.get('supervised_step', {}).get('ms_per_step'))
"
env $cfg MSPL_TRAIN_LANES=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-three-source --no-io --no-aspp --no-bs64 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('train 1 lane', d.get('train_step', {}).get('ms_per_step'))
"
done
