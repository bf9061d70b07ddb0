#!/usr/bin/env python3
"""HBM traffic of the K2 (eesp_dw_hff: tiled and direct kernels) launches from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs).

Corrections per /opt/skills/guides/MI355X_MICROARCH.md, section HBM: both counters are in KiB; on gfx950 FETCH_SIZE tallies
the 128-B requests of wide coalesced reads at 64 B, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
Usage: python tools/k2_traffic.py <dir with FETCH_SIZE_counter_collection.csv and WRITE_SIZE_...> <out.json>
The collecting commands (GPU box):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -o FETCH_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d DIR -o WRITE_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline
"""
import csv, json, sys
d, out = sys.argv[1], sys.argv[2]


FAMILY = ('eesp_dw_hff', 'eesp_dw_direct', 'eesp_dw_stream2', 'eesp_dw_exp')     # round 4: eesp_dw_exp = K2 + K3 of a stride-1 block in one launch


def k2_values(counter):
    rows = [r for r in csv.DictReader(open('%s/%s_counter_collection.csv' % (d, counter)))
            if any(k in r['Kernel_Name'] for k in FAMILY) and r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return [float(r['Counter_Value']) for r in rows], ['eesp_dw_exp' in r['Kernel_Name'] for r in rows]


(f, fused), (w, _) = k2_values('FETCH_SIZE'), k2_values('WRITE_SIZE')
per_fwd = 13                                      # EESP depthwise launches per forward (ESPDNet-UE s=2.0)
f13, w13, fused = f[:per_fwd], w[:per_fwd], fused[:per_fwd]   # the first forward's 13 launches (every forward repeats them)
read_b = [2.0 * 1024.0 * v for v in f13]
write_b = [1024.0 * v for v in w13]
res = {'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py --no-graph, batch 16 x 3 x 288 x 480',
       'corrections': 'KiB -> bytes; FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B); WRITE_SIZE exact',
       'launches_per_forward': per_fwd,
       'read_bytes_per_launch': read_b, 'write_bytes_per_launch': write_b,
       'fused_launch': fused,
       'avg_traffic_bytes_per_launch': (sum(read_b) + sum(write_b)) / per_fwd}
nf = sum(fused)
if nf:
    res['fused_avg_traffic_bytes_per_launch'] = sum(r + w_ for r, w_, f_ in zip(read_b, write_b, fused) if f_) / nf
if nf < per_fwd:
    res['standalone_avg_traffic_bytes_per_launch'] = sum(r + w_ for r, w_, f_ in zip(read_b, write_b, fused) if not f_) / (per_fwd - nf)
json.dump(res, open(out, 'w'), indent=1)
print('avg HBM traffic per K2 launch: %.2f MB (read %.2f + write %.2f)' % (res['avg_traffic_bytes_per_launch'] / 1e6,
                                                                          sum(read_b) / per_fwd / 1e6, sum(write_b) / per_fwd / 1e6))
