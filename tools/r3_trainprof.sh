#!/bin/bash
# rocprofv3 kernel trace + stats of 12 eager uest train steps (tools/run_train.py); the launch-by-launch view of the last step
# goes to gpurun_out/<tag>_trace.txt (tools/train_trace.py).  usage (on the GPU box): bash tools/r3_trainprof.sh <tag>
R=$GRAFT_REPO_ROOT
tag=${1:-r3t}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o tr --output-format csv -- python3 $R/tools/run_train.py > $R/gpurun_out/$tag.log 2>&1 || { tail -20 $R/gpurun_out/$tag.log; exit 1; }
python3 $R/tools/train_trace.py $R/gpurun_out/$tag > $R/gpurun_out/${tag}_trace.txt
cp $R/gpurun_out/$tag/*/*_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv 2>/dev/null || cp $R/gpurun_out/$tag/*_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
rm -rf $R/gpurun_out/$tag      # the raw trace is large; the two summaries stay
head -3 $R/gpurun_out/${tag}_trace.txt
