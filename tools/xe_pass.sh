#!/bin/bash
# the fused K2+K3 launch in the whole label pass: 3 launches x 2 batches in flight, and one batch in flight, on / off
for e in 1 0 1 0; do
  for f in 3 1; do
    MSPL_EESP_EXP=$e python bench.py --profile-pass --in-flight $f --steps 60 --warmup 12 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp=$e in_flight=$f', d['value'])"
  done
done
