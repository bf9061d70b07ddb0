#!/bin/bash
# GPU box: every rocprofv3 artefact the bench line refers to (kernel stats at 1 and 3 batches in flight, per-pass K2 and
# whole-pass HBM traffic from separate --pmc passes).  Outputs under gpurun_out/r03/; copy the summaries into profiles/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# the kernel set of the headline's lanes (K1 and K2 as two launches: throughput mode); one batch in flight alone would use the fused
# K1+K2 launch for the stride-1 blocks, and the K2 traffic script expects the 13 K2 launches of a forward
export MSPL_EESP_FRONT=0
rocprofv3 --kernel-trace --stats -d $O/if1 -o pp --output-format csv -- python3 $R/bench.py --profile-pass --in-flight 1 --steps 60 --warmup 10 > $O/if1.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/if3 -o pp --output-format csv -- python3 $R/bench.py --profile-pass --in-flight 3 --steps 90 --warmup 15 > $O/if3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc -o FETCH_SIZE -- python3 $R/bench.py --profile-pass --in-flight 1 --no-graph --steps 3 --warmup 1 > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc -o WRITE_SIZE -- python3 $R/bench.py --profile-pass --in-flight 1 --no-graph --steps 3 --warmup 1 > $O/pmc_w.log 2>&1
cd $R
python tools/k2_traffic.py $O/pmc $O/k2_hbm_traffic.json
python tools/pass_traffic.py $O/pmc $O/pass_hbm_traffic.json
python tools/per_kernel.py $O/if1/pp_kernel_stats.csv $O/pass_hbm_traffic.json $O/per_kernel.json | head -30
ls $O
# the uest train step (12 eager steps): kernel stats + the launch-by-launch trace of the last step
bash $R/tools/r3_trainprof.sh r03_train > /dev/null
cp $R/gpurun_out/r03_train_kernel_stats.csv $O/train_step_kernel_stats.csv
cp $R/gpurun_out/r03_train_trace.txt $O/train_step_trace.txt
ls $O
