#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_supervised.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python bench.py --no-cpu-baseline --no-three-source --no-io --no-aspp --no-bs64 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('train', d.get('train_step', {}).get('ms_per_step'), 'sup', d.get('supervised_step', {}).get('ms_per_step'))
"
