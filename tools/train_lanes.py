"""uest train step time against the number of micro-batch lanes (MSPL_TRAIN_LANES): python tools/train_lanes.py 1 2 4 8"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess
for l in sys.argv[1:]:
    code = ("import os, sys, json; sys.path.insert(0, %r); import bench; "
            "print(json.dumps({k: v for k, v in bench.train_step_rate('cuda:0').items() if k in ('ms_per_step', 'micro_batch_lanes')}))" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, MSPL_TRAIN_LANES=l), capture_output=True, text=True)
    print('lanes', l, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
