"""Four supervised iterations eager vs (1 eager + 3 graph replays): the largest state-dict differences."""
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
    nets.append(m.cuda().train())
opt, eager = None, []
for _ in range(4):
    l, _, opt = supervised.train_seg_ue_step(nets[0], x, y, crit, opt)
    eager.append(float(l))
gs = supervised.GraphedSupervisedStep(nets[1], x, y, crit)
graphed = [float(gs(x, y)[0]) for _ in range(2)]
print('loss eager', eager, 'graphed', graphed)
w = []
for (k, p), (_, q) in zip(nets[0].state_dict().items(), nets[1].state_dict().items()):
    d = (p.float() - q.float()).abs()
    w.append((float(d.max()), k))
w.sort(reverse=True)
for t in w[:6]:
    print('%.3e abs  %s' % t)
