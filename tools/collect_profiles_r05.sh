#!/bin/bash
# GPU box, round 4: every rocprofv3 artefact the bench line and DESIGN.md refer to.  Outputs under gpurun_out/r05/; copy into profiles/.
#   kernel stats of the label pass at 1 and 3 launches in flight, launch-by-launch trace of one pass, FETCH_SIZE / WRITE_SIZE passes
#   (EESP depthwise family traffic, whole-pass traffic), per-kernel table, SQ instruction counters of a pass, the fused kernel's own
#   counters, train-step and supervised-iteration kernel stats.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/if1 -o pp --output-format csv -- python3 $R/bench.py --profile-pass --in-flight 1 --steps 60 --warmup 10 > $O/if1.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/if3 -o pp --output-format csv -- python3 $R/bench.py --profile-pass --in-flight 3 --steps 90 --warmup 15 > $O/if3.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc -o FETCH_SIZE -- python3 $R/bench.py --profile-pass --in-flight 1 --no-graph --steps 3 --warmup 1 > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc -o WRITE_SIZE -- python3 $R/bench.py --profile-pass --in-flight 1 --no-graph --steps 3 --warmup 1 > $O/pmc_w.log 2>&1
echo "pmc done"
cd $R
cp $O/if1/pp_kernel_stats.csv $O/kernel_stats_inflight1.csv
cp $O/if3/pp_kernel_stats.csv $O/kernel_stats_inflight3.csv
python tools/k2_traffic.py $O/pmc $O/k2_hbm_traffic.json
python tools/pass_traffic.py $O/pmc $O/pass_hbm_traffic.json | head -20
python tools/per_kernel.py $O/kernel_stats_inflight1.csv $O/pass_hbm_traffic.json $O/per_kernel.json | head -24
bash tools/r3_passtrace.sh r05_pass > /dev/null 2>&1; cp gpurun_out/r05_pass_trace.txt $O/pass_trace.txt
bash tools/pass_pmc.sh > $O/pass_instruction_counts.txt 2>&1; tail -3 $O/pass_instruction_counts.txt
bash tools/xe_pmc.sh > $O/eesp_exp_counters.txt 2>&1; tail -4 $O/eesp_exp_counters.txt
bash tools/r3_trainprof.sh r05_train > /dev/null 2>&1
cp gpurun_out/r05_train_kernel_stats.csv $O/train_step_kernel_stats.csv; cp gpurun_out/r05_train_trace.txt $O/train_step_trace.txt
cd /tmp && rocprofv3 --kernel-trace --stats -d $O/sup -o tr --output-format csv -- python3 $R/tools/run_sup.py > $O/sup.log 2>&1; cd $R
cp $O/sup/*/*_kernel_stats.csv $O/supervised_kernel_stats.csv 2>/dev/null || cp $O/sup/*_kernel_stats.csv $O/supervised_kernel_stats.csv
rm -rf $O/sup $O/if1/*trace* $O/if3/*trace* $O/pmc/*trace*
ls $O
