#!/bin/bash
# GPU box: rebuild K2 with the in-kernel stamps and print phase timelines for tile-shape overrides.
cd $GRAFT_REPO_ROOT
touch mspl_amd/csrc/eesp_dw.hip
make -C mspl_amd/csrc STAMPS=1 > gpurun_out/k2x_build.log 2>&1 || { tail -5 gpurun_out/k2x_build.log; exit 1; }
for cfg in "A=0" "MSPL_DW_PERSIST=1" "MSPL_DW_CP=4 MSPL_DW_THREADS=576" "MSPL_DW_CP=1 MSPL_DW_THREADS=192"; do
  echo "=== $cfg"
  env MSPL_DW_STAMP=1 $cfg python tools/bench_ops.py k2x 2>&1 | grep -E "k2 stamp|^---" | head -60
done
