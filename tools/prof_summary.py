#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace/--stats CSV directory: per-kernel totals and (optionally) the
launch-by-launch trace of the last forward pass.  Usage: python tools_prof.py gpurun_out/profN [--trace]"""
import csv, glob, sys
d = sys.argv[1]
st = (glob.glob(d + '/*/*_kernel_stats.csv') + glob.glob(d + '/*_kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(st)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
tr = list(csv.DictReader(open((glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0])))
nfw = sum(1 for r in tr if 'label_epilogue' in r['Kernel_Name'])
print('total %.2f ms over %d label passes -> %.3f ms/pass' % (tot / 1e6, nfw, tot / 1e6 / max(nfw, 1)))
for r in rows[:18]:
    print('%-62s calls=%5s avg=%8.1fus %5.1f%%' % (r['Name'].replace('void mspl::', '').replace('mspl::', '')[:62], r['Calls'],
                                                  float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
if '--trace' in sys.argv:
    tr.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(tr) if 'label_epilogue' in r['Kernel_Name']]
    a, b = idx[-2] + 2, idx[-1] + 2
    t = 0
    for r in tr[a:b]:
        dd = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        t += dd
        print('%-60s grid=%8s vgpr=%4s %8.1f us' % (r['Kernel_Name'].replace('void mspl::', '').replace('mspl::', '')[:60], r['Grid_Size_X'], r['VGPR_Count'], dd))
    print('# sum %.1f us over %d launches' % (t, b - a))
