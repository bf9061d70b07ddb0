#!/usr/bin/env python3
"""Register / occupancy audit of every kernel: compiles each .hip with -Rpass-analysis=kernel-resource-usage (device only) and lists the
kernels with VGPR spills, > 20 spilled SGPRs or <= 2 waves per SIMD.  Usage: python tools/resource_audit.py [all]"""
import glob, os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mspl_amd', 'csrc')
NOSLP = {'pyrpool_sep.hip', 'eesp_dw.hip', 'pyrpool_stream.hip', 'pyrpool_train.hip', 'train.hip'}      # as in the Makefile


def one(f):
    ex = ['-fno-slp-vectorize'] if os.path.basename(f) in NOSLP else []
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', *ex,
                        '--cuda-device-only', '-c', f, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'],
                       capture_output=True, text=True, cwd=root)
    return f, r.stderr


rows = []
with ThreadPoolExecutor(8) as ex:
    for f, txt in ex.map(one, sorted(glob.glob(os.path.join(root, '*.hip')))):
        cur = None
        for l in txt.splitlines():
            m = re.search(r'remark:\s+(.*?)\s*\[-Rpass', l)
            if not m:
                continue
            t = m.group(1)
            if t.startswith('Function Name:'):
                cur = {'name': t.split(': ')[1], 'file': os.path.basename(f)}
                rows.append(cur)
            elif cur is not None:
                k, _, v = t.partition(':')
                cur[k.strip()] = v.strip()
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    r['dem'] = n.split('(')[0].replace('void mspl::', '').replace('mspl::', '')
show = rows if len(sys.argv) > 1 else [r for r in rows if int(r.get('VGPRs Spill', '0')) > 0 or int(r.get('SGPRs Spill', '0')) > 20
                                       or int(r.get('Occupancy [waves/SIMD]', '8')) <= 2]
for r in sorted(show, key=lambda r: (r['file'], r['dem'])):
    print('%-20s %-72s VGPR %4s occ %s sgpr-spill %4s vgpr-spill %3s scratch %s' % (
        r['file'], r['dem'][:72], r.get('VGPRs'), r.get('Occupancy [waves/SIMD]'), r.get('SGPRs Spill'), r.get('VGPRs Spill'),
        r.get('ScratchSize [bytes/lane]')))
print(len(rows), 'kernels,', len(show), 'listed')
