#!/bin/bash
# round 4: rocprofv3 kernel stats of the label pass, one and three launches in flight (usage: tools/r4_stats.sh TAG)
T=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/if1 -o pp --output-format csv -- python3 $R/bench.py --profile-pass --in-flight 1 --steps 60 --warmup 10 > $O/if1.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/if3 -o pp --output-format csv -- python3 $R/bench.py --profile-pass --in-flight 3 --steps 90 --warmup 15 > $O/if3.log 2>&1
cd $R
tail -1 $O/if1.log; tail -1 $O/if3.log
for f in if1 if3; do
  python - $O/$f/pp_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(sys.argv[1], 'total ms', tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:28]:
    print('%-90s %6d %9.1f us avg %5.1f %%' % (r['Name'][:90], int(r['Calls']), float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
done
