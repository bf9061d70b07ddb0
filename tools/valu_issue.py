"""Vector-issue view of the label pass: per kernel, the time its vector instructions alone need on the chip's 1 024 SIMDs
(2 cycles per wave instruction at 2.4 GHz, /opt/skills/guides/MI355X_MICROARCH.md 'Wave scheduling'; packed fp32 instructions
cost two slots, so this is a LOWER bound for the kernels hipcc packs) next to the measured time and the HBM view.
python tools/valu_issue.py profiles/r05_pass_instruction_counts.txt profiles/r05_per_kernel.json"""
import json, re, sys
counts, per = sys.argv[1], json.load(open(sys.argv[2]))
SIMDS, HZ, CYC = 1024, 2.4e9, 2
rows = {}
for line in open(counts):
    m = re.match(r'^(\S.*?)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s*$', line)
    if m:
        rows[m.group(1).strip()] = (int(m.group(2)), int(m.group(3)))
ks = per['kernels']
ks = ks if isinstance(ks, list) else [dict(kernel=k, **v) for k, v in ks.items()]
print('# %s\n# %s' % (counts, per['source']))
print('%-46s %3s %9s %9s %7s %7s' % ('kernel', 'n', 'us/pass', 'valu us', 'valu', 'hbm'))
merged = {}                                    # the counter file truncates names to 44 characters: template variants beyond that are one row
for k in ks:
    m = merged.setdefault(k['kernel'][:44], dict(kernel=k['kernel'][:44], us_per_pass=0.0, mb=0.0))
    m['us_per_pass'] += k['us_per_pass']; m['mb'] += k['hbm_MB_per_pass']
for m in merged.values():
    m['frac_of_8TBs'] = m['mb'] * 1e6 / (m['us_per_pass'] * 1e-6) / 8e12
tot_us = tot_valu = 0.0
for k in sorted(merged.values(), key=lambda k: -k['us_per_pass']):
    name = k['kernel']
    hit = [r for r in rows if name.startswith(r[:44]) or r.startswith(name[:44])]
    if not hit:
        continue
    n, valu = rows[hit[0]]
    us = valu * CYC / SIMDS / HZ * 1e6
    tot_us += k['us_per_pass']; tot_valu += us
    print('%-46s %3d %9.1f %9.1f %7.2f %7.2f' % (name[:46], n, k['us_per_pass'], us, us / k['us_per_pass'], k['frac_of_8TBs']))
print('%-46s %3s %9.1f %9.1f %7.2f' % ('TOTAL (matched kernels)', '', tot_us, tot_valu, tot_valu / tot_us))
