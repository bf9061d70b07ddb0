#!/bin/bash
# launch-by-launch trace of one label pass (one in flight) incl. runtime blit kernels.  usage (GPU box): bash tools/r3_passtrace.sh <tag> [extra bench args]
R=$GRAFT_REPO_ROOT
tag=${1:-r3p}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o tr --output-format csv -- python3 $R/bench.py --profile-pass --in-flight 1 --steps 24 --warmup 4 "$@" > $R/gpurun_out/$tag.log 2>&1 || { tail -20 $R/gpurun_out/$tag.log; exit 1; }
python3 $R/tools/prof_summary.py $R/gpurun_out/$tag --trace > $R/gpurun_out/${tag}_trace.txt
cp $R/gpurun_out/$tag/*/*_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv 2>/dev/null || cp $R/gpurun_out/$tag/*_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
rm -rf $R/gpurun_out/$tag
tail -3 $R/gpurun_out/${tag}_trace.txt
