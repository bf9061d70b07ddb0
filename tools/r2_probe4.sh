#!/bin/bash
cd $GRAFT_REPO_ROOT
MSPL_DW_DIRECT=2 timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "eesp or dw" 2>&1 | tail -3
MSPL_DW_DIRECT=2 MSPL_DW_RV=2 timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "eesp or dw" 2>&1 | tail -3
for cfg in "MSPL_DW_DIRECT=0" "MSPL_DW_DIRECT=2" "MSPL_DW_DIRECT=2 MSPL_DW_RV=2" "MSPL_DW_DIRECT=2 MSPL_DW_WT=0"; do
echo "=== $cfg"; env $cfg timeout -k 10 200 python tools/bench_ops.py k2 2>&1 | tail -5
done
