#!/bin/bash
# GPU box: the round-1 K2 kernel under the same micro-benchmark (A/B reference).
cd $GRAFT_REPO_ROOT
cp mspl_amd/csrc/eesp_dw.hip /tmp/eesp_new.hip
cp tools/alt/eesp_dw_r1.hip.txt mspl_amd/csrc/eesp_dw.hip
make -C mspl_amd/csrc > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
echo "=== round-1 kernel"
timeout -k 10 120 python tools/bench_ops.py k2 2>&1 | grep -v amdgpu.ids
cp /tmp/eesp_new.hip mspl_amd/csrc/eesp_dw.hip
