#!/bin/bash
cd $GRAFT_REPO_ROOT
python __graft_entry__.py smoke 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "=== conv1x1 default"; timeout -k 10 200 python tools/bench_ops.py conv1x1 2>&1 | tail -16
echo "=== conv1x1 NSUB=1"; MSPL_PW_NSUB=1 timeout -k 10 200 python tools/bench_ops.py conv1x1 2>&1 | tail -16
echo "=== k2"; timeout -k 10 200 python tools/bench_ops.py k2 2>&1 | tail -6
