#!/bin/bash
# GPU box: SQ counters of the fused K2+K3 kernel (isolated launches of tools/bench_ops.py expx), two --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/xepmc -o a -- python3 $R/tools/bench_ops.py expx > $R/gpurun_out/xepmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/xepmc -o b -- python3 $R/tools/bench_ops.py expx > $R/gpurun_out/xepmc_b.log 2>&1
cd $R && python - <<'PY'
import csv, collections, glob
for tag in 'ab':
    f = glob.glob('gpurun_out/xepmc/%s_counter_collection.csv' % tag)
    if not f: print('no file', tag); continue
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if 'eesp_dw_exp' in r['Kernel_Name']:
            k = 'L4' if '<128' in r['Kernel_Name'] else 'L3'
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(k, ' '.join('%s=%.3g' % (c, sorted(x)[len(x) // 2]) for c, x in sorted(v.items())))
PY
