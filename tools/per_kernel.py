#!/usr/bin/env python3
"""Per-kernel table of ONE label pass: us per pass, launches per pass, measured HBM MB per pass, fraction of the 8 TB/s roof.

Inputs: the kernel_stats CSV of
    rocprofv3 --kernel-trace --stats -d DIR -o NAME --output-format csv -- python3 bench.py --profile-pass --in-flight 1 --steps 60 --warmup 10
(--profile-pass: nothing but label passes runs, so no K2 re-issue of bench.py's roofline section is mixed in) and, optionally,
the per-kernel traffic JSON that tools/pass_traffic.py writes from the two --pmc passes.
Usage: python tools/per_kernel.py <kernel_stats.csv> [pass_traffic.json] <out.json>"""
import csv
import json
import sys

stats = sys.argv[1]
traffic = json.load(open(sys.argv[2])) if len(sys.argv) > 3 else None
out = sys.argv[-1]
rows = list(csv.DictReader(open(stats)))


def short(n):
    return n.replace('void mspl::', '').replace('mspl::', '').split('(')[0]


passes = sum(int(r['Calls']) for r in rows if 'label_epilogue' in r['Name'])
# Launches that are not part of a pass: the process's ONE-TIME parameter upload / flat-buffer set-up (`__amd_rocclr_copyBuffer`,
# a constant 882 calls whatever --steps is) and the allocator's fills.  They are listed apart, not divided by the number of passes.
ONE_TIME = ('__amd_rocclr_copyBuffer', '__amd_rocclr_fillBuffer', 'FillFunctor', 'spin_kernel')
one_time = [{'kernel': short(r['Name']), 'calls': int(r['Calls']), 'total_us': round(float(r['TotalDurationNs']) / 1e3, 1)}
            for r in rows if any(k in r['Name'] for k in ONE_TIME)]
rows = [r for r in rows if not any(k in r['Name'] for k in ONE_TIME)]
tab, tot = [], 0.0
for r in rows:
    name = short(r['Name'])
    us = float(r['TotalDurationNs']) / 1e3 / passes
    tot += us
    e = {'kernel': name, 'launches_per_pass': round(int(r['Calls']) / passes, 2), 'us_per_pass': round(us, 1),
         'avg_us': round(float(r['AverageNs']) / 1e3, 2)}
    if traffic is not None:
        t = traffic['per_kernel'].get(name)
        if t is not None:
            mb = t['read_MB'] + t['write_MB']
            e['hbm_MB_per_pass'] = round(mb, 1)
            e['frac_of_8TBs'] = round(mb * 1e6 / (us * 1e-6) / 8e12, 3) if us > 0 else None
    tab.append(e)
tab.sort(key=lambda e: -e['us_per_pass'])
for e in tab:
    e['share'] = round(e['us_per_pass'] / tot, 3)
res = {'source': 'rocprofv3 --kernel-trace --stats of bench.py --profile-pass --in-flight 1 (bs16, 16x3x288x480), %d passes' % passes,
       'kernel_us_per_pass': round(tot, 1), 'kernels': tab[:24],
       'not_in_a_pass': {'note': 'one-time set-up launches of the process (state-dict upload, fills): excluded from the per-pass table', 'rows': one_time}}
json.dump(res, open(out, 'w'), indent=1)
print('kernel time per pass: %.1f us over %d passes' % (tot, passes))
for e in tab[:20]:
    print('  %-44s %6.1f us  x%-5.2f %5.1f%%  %s' % (e['kernel'][:44], e['us_per_pass'], e['launches_per_pass'], 100 * e['share'],
                                                   ('%.0f MB %.2f' % (e['hbm_MB_per_pass'], e['frac_of_8TBs'])) if 'hbm_MB_per_pass' in e else ''))
