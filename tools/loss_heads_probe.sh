#!/bin/bash
# kernel times of tools/loss_heads_probe.py (rocprofv3 --kernel-trace): the one-launch loss against the five launches it replaces
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05/lh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o lh --output-format csv -- python3 $R/tools/loss_heads_probe.py > $O/run.log 2>&1
cd $R
python - $O/lh_kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = {}
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void mspl::', '').replace('mspl::', '')[:60]
    out.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in out.items():
    print('%-62s n=%3d  first five %s  last five %s' % (k, len(v), [round(x, 1) for x in v[:5]], [round(x, 1) for x in v[-5:]]))
PY
