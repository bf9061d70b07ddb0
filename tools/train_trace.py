#!/usr/bin/env python3
"""Launch-by-launch view of ONE eager uest train step from a rocprofv3 --kernel-trace CSV (tools/r3_trainprof.sh):
the launches between the last two adam_kernel launches, in start order, with durations, grid sizes and the running sum.
Usage: python tools/train_trace.py gpurun_out/r3t [--all]"""
import csv, glob, re, sys
d = sys.argv[1]
f = (glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0]
tr = list(csv.DictReader(open(f)))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
t = 0.0
fwd_end = None
rows = []
for r in tr[a:b]:
    dd = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    t += dd
    n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('mspl::', '')
    n = re.sub(r'at::native::', 'aten::', n)[:70]
    rows.append((n, dd, r['Grid_Size_X'], r.get('Workgroup_Size_X', '')))
print('# %d launches, %.1f us of kernel time' % (len(rows), t))
acc = 0.0
for n, dd, g, wg in rows:
    acc += dd
    print('%-70s grid=%9s wg=%4s %8.1f us  (cum %8.1f)' % (n, g, wg, dd, acc))
