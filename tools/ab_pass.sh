#!/bin/bash
# A/B of one environment switch in the whole label pass: usage tools/ab_pass.sh VAR VALUE_A VALUE_B  (three launches x 2 batches in flight, and one)
V=$1; A=$2; B=$3
for e in $A $B $A $B; do
  for f in 3 1; do
    env $V=$e python bench.py --profile-pass --in-flight $f --steps 60 --warmup 12 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$e in_flight=$f', d['value'])"
  done
done
