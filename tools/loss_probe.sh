#!/bin/bash
R=$GRAFT_REPO_ROOT
for b in "$@"; do
O=$R/gpurun_out/lp_$b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MSPL_LOSS_BLOCKS=$b rocprofv3 --kernel-trace --stats -d $O -o lp --output-format csv -- python3 $R/tools/loss_probe.py > $O/run.log 2>&1
cd $R
python - $O/lp_kernel_trace.csv $b <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = {}
for r in rows:
    for k in ('uw_loss_kernel', 'wce_fwd_kernel', 'wce_bwd_kernel'):
        if k in r['Kernel_Name']:
            out.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('blocks', sys.argv[2], {k: [round(x, 1) for x in v[-7:]] for k, v in out.items()})
PY
done
