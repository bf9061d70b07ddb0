"""Same-box A/B of the uest train step (bench.train_step_rate) over the values of one environment variable, alternating:
python tools/ab_train.py MSPL_LOSS_HEADS 0 1 [--rounds 3] [--sup]      (--sup: the supervised iteration instead)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--rounds"]
var, vals = args[0], args[1:]
rounds = int(sys.argv[sys.argv.index('--rounds') + 1]) if '--rounds' in sys.argv else 3
fn = 'supervised_step_rate' if '--sup' in sys.argv else 'train_step_rate'
code = ("import sys, json; sys.path.insert(0, %r); import bench; r = bench.%s('cuda:0'); "
        "print(json.dumps({'ms': r.get('ms_per_step')}))" % (root, fn))
res = {v: [] for v in vals}
for _ in range(rounds):
    for v in vals:
        out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, **{var: v}), capture_output=True, text=True)
        try:
            res[v].append(json.loads(out.stdout.strip().splitlines()[-1])['ms'])
        except Exception:
            print(out.stderr[-400:])
        print(var, v, res[v][-1:], flush=True)
for v in vals:
    print('%s=%s: %s  median %.3f ms' % (var, v, ['%.3f' % m for m in res[v]], sorted(res[v])[len(res[v]) // 2] if res[v] else float('nan')))
