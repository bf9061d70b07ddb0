#!/usr/bin/env python3
"""Which Python call sites issue the ATen / copy launches of one eager uest train step: torch.profiler with stacks, events that
launched a device kernel whose name is not one of ours, grouped by (op, innermost mspl_amd frame)."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from mspl_amd import models, training
from tests.synth import synth_state_dict
a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
x = torch.randn(16, 3, 256, 480, device='cuda')
tgt = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
tgt.load_state_dict(synth_state_dict(tgt.state_dict(), 9))
tgt = tgt.cuda().eval()
y = torch.randint(0, 5, (16, 256, 480), device='cuda')
cw = torch.ones(5)
loss, opt = training.train_step(tgt, x, y, cw, None, ignore_idx=4)
for _ in range(2):
    loss, opt = training.train_step(tgt, x, y, cw, opt, ignore_idx=4)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    loss, opt = training.train_step(tgt, x, y, cw, opt, ignore_idx=4)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0:
        continue
    if any(k.name.startswith('aten::') for k in ev.cpu_children if k.device_time_total > 0):
        continue                                    # count the innermost op that owns the kernel
    site = ''
    for fr in ev.stack:
        if 'mspl_amd' in fr or 'tools/' in fr:
            site = fr.split('/')[-1]
            break
    if not site:
        p, names = ev.cpu_parent, []
        while p is not None and len(names) < 3:
            names.append(p.name)
            p = p.cpu_parent
        site = ' < '.join(names)
    shp = str(ev.input_shapes)[:60]
    k = (ev.name, site[:110], shp)
    agg[k][0] += 1
    agg[k][1] += ev.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print('ATen ops with device time in one eager train step: %d launches, %.1f us' % (sum(v[0] for v in agg.values()), tot))
for (name, site, shp), (n, t) in rows[:70]:
    print('%7.1f us %4d  %-22s %-60s %s' % (t, n, name, shp, site))
