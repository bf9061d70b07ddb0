#!/bin/bash
# GPU box: tile-shape sweep of the tiled K2 kernel at the batch the lanes launch (32), stride-1 shapes of levels 3 / 4.
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env K2_N=32 "$@" python tools/bench_ops.py k2 2>/dev/null | sed -n '3p;5p'; }
run A=0
run MSPL_DW_PERSIST=1
run MSPL_DW_PERSIST=0
for cp in 1 2 4; do for th in 3 6 9 18; do for t in 128 192 256; do run MSPL_DW_CP=$cp MSPL_DW_TH=$th MSPL_DW_THREADS=$t; done; done; done
