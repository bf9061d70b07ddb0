#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "MSPL_PW_PIPE_LDS=40" "MSPL_PW_PIPE_LDS=72" "MSPL_PW_PIPE_LDS=72 MSPL_PW_NSUB=1" "MSPL_PW_PIPE_LDS=72 MSPL_PW_TPW=2" "MSPL_PW_PIPE_LDS=40 MSPL_PW_TPW=2"; do
echo "=== $cfg"; env $cfg timeout -k 10 200 python tools/bench_ops.py conv1x1 2>&1 | grep -E "L3 exp|L4 exp|L4_0 exp|L3_0 exp"
done
