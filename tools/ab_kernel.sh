#!/bin/bash
# Per-kernel A/B of two builds of the library inside the label pass (one launch in flight): usage tools/ab_kernel.sh PATTERN LIB_A LIB_B
# prints the rocprofv3 --stats rows of the kernels matching PATTERN for each library.
P=$1; shift
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  export MSPL_HIP_LIB=$L
  rm -rf /tmp/abk && rocprofv3 --kernel-trace --stats -d /tmp/abk -o abk --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --profile-pass --in-flight 1 --steps 30 --warmup 6 > /dev/null 2>&1
  echo "== $L"
  f=$(find /tmp/abk -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$P" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r['Name']):
        print('%-72s calls %5s  avg %9.2f us  total %10.1f us' % (r['Name'][6:78], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
PY
done
