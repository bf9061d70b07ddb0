#!/usr/bin/env python3
"""Which ATen ops (not mspl kernels) run inside one eager uest train step: torch.profiler table grouped by op."""
import argparse, os, sys
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    loss, opt = training.train_step(tgt, x, y, cw, opt, ignore_idx=4)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=45, max_name_column_width=60))
