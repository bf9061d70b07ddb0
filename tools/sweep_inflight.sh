# lanes x batches-per-launch sweep of the label pass (GPU box): bash tools/sweep_inflight.sh
for cfg in "3 2" "4 2" "3 3" "2 3" "4 1" "4 3" "3 2"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --no-io --no-aspp --no-three-source --no-train --no-bs64 --in-flight $1 --group $2 --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('in-flight $1 group $2:', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'))"
done
