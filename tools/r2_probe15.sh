#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== stream"; timeout -k 10 120 python tools/c3_probe.py 2>&1 | grep -v amdgpu
echo "== tiled"; MSPL_GC3S=0 timeout -k 10 120 python tools/c3_probe.py 2>&1 | grep -v amdgpu
