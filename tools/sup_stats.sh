#!/bin/bash
# rocprofv3 kernel stats of 8 eager supervised iterations (tools/run_sup.py); usage: tools/sup_stats.sh TAG
T=${1:-sup}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o sup --output-format csv -- python3 $R/tools/run_sup.py > $O/run.log 2>&1
cd $R
python - $O/sup_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total ms per iteration', tot / 1e6 / 8, 'launches per iteration', sum(int(r['Calls']) for r in rows) / 8)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:40]:
    print('%-80s %6.1f/it %8.1f us avg %7.1f us/it %5.1f %%' % (r['Name'].replace('void mspl::','')[:80], int(r['Calls']) / 8, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 8e3, 100 * float(r['TotalDurationNs']) / tot))
PY
