"""Per-kernel average duration with one and with three label-pass launches in flight (rocprofv3 --kernel-trace --stats of
`bench.py --profile-pass --in-flight 1|3`): which kernels stretch most when the passes share the chip, and each kernel's share
of all kernel time at three in flight.
python tools/inflight_contention.py profiles/r05_kernel_stats_inflight1.csv profiles/r05_kernel_stats_inflight3.csv"""
import csv, sys


def load(p):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(p))}


a, b = load(sys.argv[1]), load(sys.argv[2])
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print('# %s -> %s' % (sys.argv[1], sys.argv[2]))
print('# kernel time of the run: %.1f ms (one in flight), %.1f ms (three in flight)' % (ta / 1e6, tb / 1e6))
print('%-58s %6s %9s %9s %7s %7s' % ('kernel', 'share3', 'us (1)', 'us (3)', 'ratio', 'calls'))
rows = []
for k in a:
    if k in b and k.startswith(('void mspl::', 'mspl::')):
        pa, pb = a[k][1] / a[k][0], b[k][1] / b[k][0]
        name = k.replace('void ', '').replace('mspl::', '')
        rows.append((b[k][1] / tb, name[:58], pa / 1e3, pb / 1e3, pb / pa, b[k][0]))
for r in sorted(rows, reverse=True):
    print('%-58s %6.3f %9.1f %9.1f %7.2f %7d' % (r[1], r[0], r[2], r[3], r[4], r[5]))
