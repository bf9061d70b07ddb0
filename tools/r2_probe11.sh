#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "32 3" "32 4" "48 2" "48 3" "64 2" "64 3" "96 2" "128 1" "128 2"; do
set -- $cfg
echo -n "batch $1 lanes $2: "
timeout -k 10 160 python bench.py --profile-pass --profile-batch $1 --in-flight $2 --steps 60 --warmup 10 2>&1 | grep -o '"value": [0-9.]*'
done
