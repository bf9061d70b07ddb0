#!/usr/bin/env python3
"""HBM traffic of ONE whole label pass (every kernel) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs),
corrected like tools/k2_traffic.py (KiB -> bytes; FETCH_SIZE x2 on gfx950; WRITE_SIZE exact).  FETCH_SIZE x2 over-counts narrow
reads (only 128-B requests are tallied at half), so the read figure is an upper bound for kernels with scalar / 4-byte loads.
Usage: python tools/pass_traffic.py <dir> <out.json>; the collecting commands (GPU box):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -o FETCH_SIZE -- python3 bench.py --steps 2 --warmup 1 --in-flight 1 --no-graph --no-cpu-baseline --no-train --no-aspp --no-io --no-bs64
  (same with WRITE_SIZE)"""
import collections, csv, json, sys
d, out = sys.argv[1], sys.argv[2]


def per_kernel(counter):
    rows = [r for r in csv.DictReader(open('%s/%s_counter_collection.csv' % (d, counter))) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    marks = [i for i, r in enumerate(rows) if 'label_epilogue' in r['Kernel_Name']]
    a, b = marks[1] + 2, marks[2] + 2                  # one steady-state pass: after the epilogue+merge of pass 1 .. pass 2's merge
    agg = collections.OrderedDict()
    for r in rows[a:b]:
        k = r['Kernel_Name'].replace('void mspl::', '').replace('mspl::', '').split('(')[0]
        agg[k] = agg.get(k, 0.0) + float(r['Counter_Value'])
    return agg


f, w = per_kernel('FETCH_SIZE'), per_kernel('WRITE_SIZE')
tab = {k: {'read_MB': 2.0 * 1024 * f.get(k, 0) / 1e6, 'write_MB': 1024 * w.get(k, 0) / 1e6} for k in set(f) | set(w)}
tot_r, tot_w = sum(v['read_MB'] for v in tab.values()), sum(v['write_MB'] for v in tab.values())
res = {'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), one label pass of bench.py --no-graph, batch 16 x 3 x 288 x 480',
       'corrections': 'KiB -> bytes; FETCH_SIZE x2 (upper bound for narrow loads); WRITE_SIZE exact',
       'read_MB_per_pass': tot_r, 'write_MB_per_pass': tot_w, 'total_MB_per_pass': tot_r + tot_w,
       'total_MB_per_image': (tot_r + tot_w) / 16, 'per_kernel': tab}
json.dump(res, open(out, 'w'), indent=1)
print('HBM traffic per pass: read %.0f MB + write %.0f MB = %.0f MB (%.1f MB/image)' % (tot_r, tot_w, tot_r + tot_w, (tot_r + tot_w) / 16))
for k, v in sorted(tab.items(), key=lambda kv: -(kv[1]['read_MB'] + kv[1]['write_MB']))[:14]:
    print('  %-40s read %8.1f  write %8.1f MB' % (k[:40], v['read_MB'], v['write_MB']))
