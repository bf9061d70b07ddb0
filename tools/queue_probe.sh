#!/bin/bash
# Does GPU_MAX_HW_QUEUES change the lanes' overlap?  usage: tools/queue_probe.sh  (train step x 5 and label pass x 2 per setting)
for q in default 8; do
  for i in 1 2 3 4 5; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    python -c "
import bench, torch
r = bench.train_step_rate(torch.device('cuda:0')); print('queues $q train', r['ms_per_step'], r['ms_per_step_min_max'])
" 2>&1 | tail -1
  done
  for i in 1 2; do
    python bench.py --profile-pass --in-flight 3 --steps 60 --warmup 12 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q pass', d['value'])"
  done
done
