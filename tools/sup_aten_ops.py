#!/usr/bin/env python3
"""Which ATen ops run inside one eager supervised iteration (torch.profiler, grouped by op)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from mspl_amd import models, supervised, losses
from tests.synth import synth_state_dict
a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
x = torch.randn(16, 3, 288, 480, device='cuda')
m = models.ESPDNetwithUncertaintyEstimation(a, classes=13, dataset='camvid', fix_pyr_plane_proj=True)
m.load_state_dict(synth_state_dict(m.state_dict(), 9))
m = m.cuda().train()
y = torch.randint(0, 13, (16, 288, 480), device='cuda')
crit = losses.SegmentationLoss(n_classes=13, device='cuda', ignore_idx=255)
loss, _, opt = supervised.train_seg_ue_step(m, x, y, crit, None)
for _ in range(2):
    loss, _, opt = supervised.train_seg_ue_step(m, x, y, crit, opt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    loss, _, opt = supervised.train_seg_ue_step(m, x, y, crit, opt)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=60, max_name_column_width=60))
