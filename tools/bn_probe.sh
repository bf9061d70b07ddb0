#!/bin/bash
# usage: tools/bn_probe.sh TAG  (env passes through)
T=${1:-bnp}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o bn --output-format csv -- python3 $R/tools/bn_probe.py > $O/run.log 2>&1
cd $R
python - $O/bn_kernel_trace.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
keys = ('bn_stats_fused', 'affine_prelu_bwd', 'pointwise_kernel', 'bn_train_bwd_apply')
seq = [(k, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // 256) for r in rows for k in keys if k in r['Kernel_Name']]
# per shape: 5 reps x (stats, pointwise, bwd, pointwise)
for s in range(len(seq) // 20):
    blk = seq[s * 20:(s + 1) * 20]
    last = blk[16:20]
    print('shape %d: ' % s + '  '.join('%s %.1f us (%d wg)' % (k[:12], d, g) for k, d, g in last))
PY
