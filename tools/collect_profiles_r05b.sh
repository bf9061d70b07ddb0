#!/bin/bash
# GPU box, second half of round 5 (training kernels changed, label-pass kernels did not): train-step and supervised kernel stats, the
# per-step trace, the per-kernel traffic table, the graphed lanes' per-kernel totals, the weight-gradient and loss probes.
# Outputs under gpurun_out/r05/; copy into profiles/ as r05_*.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
bash tools/r3_trainprof.sh r05_train > /dev/null 2>&1
cp gpurun_out/r05_train_kernel_stats.csv $O/train_step_kernel_stats.csv; cp gpurun_out/r05_train_trace.txt $O/train_step_trace.txt
head -1 $O/train_step_trace.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/sup -o tr --output-format csv -- python3 $R/tools/run_sup.py > $O/sup.log 2>&1; cd $R
cp $O/sup/*/*_kernel_stats.csv $O/supervised_kernel_stats.csv 2>/dev/null || cp $O/sup/*_kernel_stats.csv $O/supervised_kernel_stats.csv
rm -rf $O/sup
bash tools/train_traffic.sh > $O/train_step_traffic.txt 2>&1; head -3 $O/train_step_traffic.txt
rm -rf $O/tt
bash tools/train_lanes_prof.sh > $O/train_lanes_kernel_totals.txt 2>&1; head -2 $O/train_lanes_kernel_totals.txt
rm -rf $O/tl
python tools/wgrad_probe.py 2>&1 | grep -v amdgpu.ids > $O/wgrad_probe.txt; tail -1 $O/wgrad_probe.txt
bash tools/loss_heads_probe.sh 2>&1 | grep "uw_loss\|bilinear" > $O/loss_heads_probe.txt; head -2 $O/loss_heads_probe.txt
rm -rf $O/lh
ls $O
