#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "A=0" "MSPL_PW_PIPE_LDS=72" "MSPL_PW_TPW=2" "MSPL_PW_PIPE_LDS=72 MSPL_PW_TPW=2" "MSPL_DW_DIRECT=0" "MSPL_DW_DIRECT=2" "MSPL_DW_WT=0" "MSPL_PREP_WAVES=8" "MSPL_PYR_SEG=12" "MSPL_PYR_SEG=19"; do
echo -n "$cfg: "; env $cfg timeout -k 10 160 python bench.py --profile-pass --in-flight 3 --group 2 --steps 90 --warmup 18 2>&1 | grep -o '"value": [0-9.]*'
done
