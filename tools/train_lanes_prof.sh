#!/bin/bash
# per-kernel totals of the graphed (micro-batch lanes) uest train step: sum of durations per kernel over the last replays (the lanes
# overlap: the sums add up to more than the step's wall time)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05/tl
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o tl --output-format csv -- python3 $R/tools/train_lanes_prof.py > $O/run.log 2>&1
cd $R
python - $O/tl_kernel_trace.csv <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-5] + 1, idx[-1] + 1            # the last four steps
agg = {}
for r in rows[a:b]:
    n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('mspl::', '')[:64]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    e = agg.setdefault(n, [0, 0.0]); e[0] += 1; e[1] += d
wall = (int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3 / 4
tot = sum(v[1] for v in agg.values()) / 4
print('# per step: %d launches, %.0f us of kernel durations (overlapped lanes), wall %.0f us' % (sum(v[0] for v in agg.values()) / 4, tot, wall))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%8.1f us  %4d x %6.1f  %s' % (d / 4, c / 4, d / c, n))
PY
