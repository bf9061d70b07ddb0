#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2t2 -o tr --output-format csv -- python3 $R/tools/run_train.py > $R/gpurun_out/r2t2.log 2>&1
ls $R/gpurun_out/r2t2
