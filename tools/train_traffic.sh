#!/bin/bash
# Memory traffic per kernel of ONE eager uest train step (tools/run_train.py): two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
# runs), corrected as tools/pass_traffic.py does, with the kernels' durations from the same runs' kernel traces: who moves how many bytes
# at which rate.  usage (GPU box): bash tools/train_traffic.sh [script.py]
R=$GRAFT_REPO_ROOT
S=${1:-tools/run_train.py}
O=$R/gpurun_out/r05/tt
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o $c -- python3 $R/$S > $O/$c.log 2>&1 || { tail -5 $O/$c.log; exit 1; }
done
cd $R
python - $O <<'PY'
import csv, sys, re, collections
d = sys.argv[1]
def load(counter):
    rows = [r for r in csv.DictReader(open('%s/%s_counter_collection.csv' % (d, counter))) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    tr = {int(r['Dispatch_Id']): (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open('%s/%s_kernel_trace.csv' % (d, counter)))}
    marks = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name'] or 'sgd' in r['Kernel_Name'].lower()]
    a, b = marks[-2] + 1, marks[-1] + 1
    agg = collections.OrderedDict()
    for r in rows[a:b]:
        k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('mspl::', '')[:60]
        e = agg.setdefault(k, [0, 0.0, 0.0])
        e[0] += 1; e[1] += float(r['Counter_Value']); e[2] += tr.get(int(r['Dispatch_Id']), 0.0)
    return agg
f, w = load('FETCH_SIZE'), load('WRITE_SIZE')
tab = []
for k in f:
    rd = 2.0 * 1024 * f[k][1] / 1e6; wr = 1024 * w.get(k, [0, 0, 0])[1] / 1e6
    us = 0.5 * (f[k][2] + w.get(k, [0, 0, f[k][2]])[2])
    tab.append((us, f[k][0], rd, wr, k))
tot = sum(t[0] for t in tab)
print('# one step: %.0f us of kernel time (under the counters), read %.0f MB (FETCH_SIZE x 2: upper bound for narrow loads) + write %.0f MB' % (tot, sum(t[2] for t in tab), sum(t[3] for t in tab)))
print('#      us  launches   read MB  write MB    TB/s  kernel')
for us, n, rd, wr, k in sorted(tab, reverse=True)[:45]:
    print('%9.1f %6d %10.1f %9.1f %7.2f  %s' % (us, n, rd, wr, (rd + wr) / max(us, 1e-9), k))
PY
