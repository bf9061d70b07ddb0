#!/bin/bash
# bn_train_small_fwd / _bwd per tensor shape (tools/bn_probe.py's node), from a rocprofv3 kernel trace
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05/bns
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o bn --output-format csv -- python3 $R/tools/bn_probe.py > $O/run.log 2>&1
cd $R
python - $O/bn_kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = []
for r in rows:
    n = r['Kernel_Name']
    if 'bn_' in n or 'affine_prelu' in n or 'pointwise' in n:
        out.append((n.split('(')[0].replace('void mspl::', '').replace('mspl::', '')[:40], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Grid_Size_X']), int(r['Workgroup_Size_X'])))
prev = None
for k in out:
    print('%-42s %7.1f us  grid %8d wg %4d' % k)
PY
