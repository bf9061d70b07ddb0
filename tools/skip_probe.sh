#!/bin/bash
cd $GRAFT_REPO_ROOT
for spec in none pw_l4 pw_l3 pw_rest k2_l4 k2_l3 k2_rest pyr prep c3 pool bil label pw_l4,k2_l4 pw_l4,k2_l4,pw_l3,k2_l3; do
  for d in 1 3; do
    echo -n "$spec depth=$d: "
    timeout -k 10 120 python tools/skip_probe.py $spec $d 2>&1 | grep -o '"value": [0-9.]*'
  done
done
