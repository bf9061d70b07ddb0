#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "MSPL_WGRAD_BATCH=16" "MSPL_WGRAD_BATCH=40" "MSPL_WGRAD_BATCH=200" "MSPL_WGRAD_BATCH=8"; do
echo "== $cfg"
env $cfg timeout -k 10 300 python bench.py --no-cpu-baseline --no-three-source --no-io --no-aspp --no-bs64 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('train', d.get('train_step', {}).get('ms_per_step'), 'sup', d.get('supervised_step', {}).get('ms_per_step'))
"
done
