#!/bin/bash
# marginal cost of kernel families: usage skip_probe.sh [group]
cd $GRAFT_REPO_ROOT
G=${1:-1}
for spec in none pw_l4 pw_l3 pw_rest k2_l4 k2_l3 k2_rest pyr prep c3 pool bil label pw_l4,k2_l4,pw_l3,k2_l3; do
  for d in 3; do
    echo -n "$spec depth=$d group=$G: "
    timeout -k 10 120 python tools/skip_probe.py $spec $d $G 2>&1 | grep -o '"value": [0-9.]*'
  done
done
