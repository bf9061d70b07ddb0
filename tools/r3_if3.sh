#!/bin/bash
# kernel stats of the label pass with 3 launches in flight (throughput mode).  usage (GPU box): bash tools/r3_if3.sh <tag>
R=$GRAFT_REPO_ROOT
tag=${1:-r3if3}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o pp --output-format csv -- python3 $R/bench.py --profile-pass --in-flight 3 --steps 90 --warmup 15 > $R/gpurun_out/$tag.log 2>&1 || { tail -20 $R/gpurun_out/$tag.log; exit 1; }
cp $R/gpurun_out/$tag/*/*_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv 2>/dev/null || cp $R/gpurun_out/$tag/*_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
rm -rf $R/gpurun_out/$tag
tail -2 $R/gpurun_out/$tag.log
