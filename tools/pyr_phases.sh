#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "A=0" "MSPL_PYR_STOP=1" "MSPL_PYR_STOP=2" "MSPL_PYR_STOP=3" "MSPL_PYR_STOP=257" "MSPL_PYR_STOP=1281" "MSPL_PYR_CPB=16" "MSPL_PYR_CPB=4" "MSPL_PYR_CPB=1"; do
  echo "=== $cfg"
  env $cfg timeout -k 10 120 python tools/bench_ops.py pyr 2>&1 | grep -E "kernel only|whole"
done
