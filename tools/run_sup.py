"""8 eager supervised iterations (train_seg_ue body, batch-statistics BatchNorm) for rocprofv3 --kernel-trace --stats."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import losses, models, supervised
from tests.synth import synth_state_dict
a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
m = models.ESPDNetwithUncertaintyEstimation(a, classes=13, dataset='camvid', fix_pyr_plane_proj=True)
m.load_state_dict(synth_state_dict(m.state_dict(), 10))
m = m.cuda().train()
g = torch.Generator().manual_seed(8)
x = torch.randn((16, 3, 288, 480), generator=g).cuda()
y = torch.randint(0, 13, (16, 288, 480), generator=g).cuda()
crit = losses.SegmentationLoss(n_classes=13, device='cuda', ignore_idx=255)
opt = None
for _ in range(8):
    _, _, opt = supervised.train_seg_ue_step(m, x, y, crit, opt, None, None, 1.0, 0.009, 10.0, 0.9, 4e-5, supervised.FLOOD_LEVEL)
torch.cuda.synchronize()
