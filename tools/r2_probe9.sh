#!/bin/bash
cd $GRAFT_REPO_ROOT
for l in 8 16; do
echo "== lanes $l"
MSPL_TRAIN_LANES=$l timeout -k 10 300 python bench.py --no-cpu-baseline --no-three-source --no-io --no-aspp --no-bs64 --steps 20 --warmup 5 2>gpurun_out/probe9_$l.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('train', d.get('train_step', {}).get('ms_per_step'), d.get('train_step', {}).get('loss_finite'))
"
tail -3 gpurun_out/probe9_$l.err | grep -v amdgpu.ids
done
