#!/bin/bash
# GPU box: SQ instruction counters of every kernel of one label pass (eager launches, batch 16), aggregated per kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/passpmc -o a -- python3 $R/bench.py --profile-pass --in-flight 1 --no-graph --steps 3 --warmup 1 > $R/gpurun_out/passpmc_a.log 2>&1
cd $R && python - <<'PY'
import csv, collections, glob
rows = list(csv.DictReader(open(glob.glob('gpurun_out/passpmc/a_counter_collection.csv')[0])))
rows.sort(key=lambda r: int(r['Dispatch_Id']))
marks = [int(r['Dispatch_Id']) for r in rows if 'label_epilogue' in r['Kernel_Name'] and r['Counter_Name'] == 'SQ_WAVES']
lo, hi = marks[-2], marks[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    d = int(r['Dispatch_Id'])
    if lo < d <= hi:
        k = r['Kernel_Name'].replace('void mspl::', '').replace('mspl::', '').split('(')[0][:44]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        agg[k]['n'] += 1.0 / 8
tot = collections.defaultdict(float)
print('%-44s %5s %10s %10s %9s %9s %9s' % ('kernel', 'n', 'VALU', 'SALU', 'LDS', 'VMEM_RD', 'VMEM_WR'))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_INSTS_VALU']):
    print('%-44s %5.0f %10.0f %10.0f %9.0f %9.0f %9.0f' % (k, v['n'], v['SQ_INSTS_VALU'], v['SQ_INSTS_SALU'], v['SQ_INSTS_LDS'], v['SQ_INSTS_VMEM_RD'], v['SQ_INSTS_VMEM_WR']))
    for c in v: tot[c] += v[c]
print('TOTAL VALU %.0f  SALU %.0f  LDS %.0f  VMEM_RD %.0f VMEM_WR %.0f  (wave-instructions per pass of 16 images)' % (tot['SQ_INSTS_VALU'], tot['SQ_INSTS_SALU'], tot['SQ_INSTS_LDS'], tot['SQ_INSTS_VMEM_RD'], tot['SQ_INSTS_VMEM_WR']))
PY
