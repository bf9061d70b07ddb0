#!/bin/bash
# GPU box: SQ counters of the K2 launches (eager, batch K2_N) -- two --pmc passes, kernel-trace only.
cd /tmp && export TMPDIR=/tmp
export K2_N=${K2_N:-64}
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/k2pmc -o a -- python3 $R/tools/bench_ops.py k2x > $R/gpurun_out/k2pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/k2pmc -o b -- python3 $R/tools/bench_ops.py k2x > $R/gpurun_out/k2pmc_b.log 2>&1
cd $R && python - <<'PY'
import csv, collections, glob
for tag in ('a', 'b'):
    f = glob.glob('gpurun_out/k2pmc/%s_counter_collection.csv' % tag)
    if not f:
        print('no csv for', tag, glob.glob('gpurun_out/k2pmc/*')); continue
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.OrderedDict()
    for r in rows:
        if 'eesp_dw' not in r['Kernel_Name']: continue
        key = (r['Kernel_Name'].split('(')[0][-60:], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X', ''), r['Counter_Name'])
        agg.setdefault(key, []).append(float(r['Counter_Value']))
    last = None
    for (k, gsz, c), v in agg.items():
        if (k, gsz) != last:
            print('---', k, 'grid', gsz); last = (k, gsz)
        print('   %-24s %14.0f  (n=%d)' % (c, v[-1], len(v)))
PY
