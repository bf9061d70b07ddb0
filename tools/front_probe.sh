#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "eesp" 2>&1 | tail -12 | cut -c1-160
timeout -k 10 200 python tools/front_probe.py 2>&1 | grep -v amdgpu | tail -3
for cfg in "MSPL_EESP_FRONT=1" "MSPL_EESP_FRONT=0"; do
for d in "1 1" "3 2"; do set -- $d; echo -n "$cfg depth=$1 group=$2: "; env $cfg timeout -k 10 120 python bench.py --profile-pass --in-flight $1 --group $2 --steps 90 --warmup 18 2>&1 | grep -o '"value": [0-9.]*'; done
done
