"""Two supervised iterations from the same state: which parameter gradients differ between the runs, and by how much (float atomics order only?)."""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import losses, models, supervised, autograd as ag, layers
from tests.synth import synth_state_dict, synth_input, synth_labels
H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 96
a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
x = synth_input((2, 3, H, W), 28).cuda()
y = synth_labels((2, H, W), 13, 28).cuda()
crit = losses.SegmentationLoss(n_classes=13, device='cuda', ignore_idx=255)
res = []
for rep in range(3):
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=13, dataset='camvid', fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(m.state_dict(), 5))
    m = m.cuda().train()
    with torch.enable_grad():
        out = m(x)
        loss = crit(out[0] + 0.5 * out[1], y).mean()
        loss.backward()
    res.append((float(loss), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, {k: v.clone() for k, v in m.named_buffers()}))
print('loss', [r[0] for r in res])
worst = []
for k in res[0][1]:
    g0 = res[0][1][k]
    for r in res[1:]:
        d = float((r[1][k] - g0).abs().max()); s = float(g0.abs().max()) + 1e-30
        worst.append((d / s, d, k))
worst.sort(reverse=True)
for w_ in worst[:15]:
    print('%.3e rel  %.3e abs  %s' % w_)
wb = []
for k in res[0][2]:
    b0 = res[0][2][k].float()
    for r in res[1:]:
        d = float((r[2][k].float() - b0).abs().max()); s = float(b0.abs().max()) + 1e-30
        wb.append((d / s, d, k))
wb.sort(reverse=True)
print('buffers:')
for w_ in wb[:6]:
    print('%.3e rel  %.3e abs  %s' % w_)
