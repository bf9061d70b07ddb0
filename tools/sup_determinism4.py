"""Four eager supervised iterations on two identically initialised nets: how far apart do run-to-run float-atomics differences drive the weights?"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import losses, models, supervised
from tests.synth import synth_state_dict, synth_input, synth_labels
H, W = int(sys.argv[1]), int(sys.argv[2])
a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
x = synth_input((2, 3, H, W), 28).cuda()
y = synth_labels((2, H, W), 13, 28).cuda()
crit = losses.SegmentationLoss(n_classes=13, device='cuda', ignore_idx=255)
nets = []
for _ in range(2):
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=13, dataset='camvid', fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(m.state_dict(), 5))
    m = m.cuda().train()
    opt = None
    for _ in range(4):
        l, _, opt = supervised.train_seg_ue_step(m, x, y, crit, opt)
    nets.append((float(l), {k: v.clone() for k, v in m.state_dict().items()}))
print('loss', nets[0][0], nets[1][0])
w = []
for k, p in nets[0][1].items():
    q = nets[1][1][k]
    d = (p.float() - q.float()).abs()
    w.append((float(d.max()), float((d / (q.float().abs() + 1e-12)).max()), k))
w.sort(reverse=True)
for t in w[:8]:
    print('%.3e abs  %.3e rel  %s' % t)
