"""Throughput of the 3-source label pass (BASELINE configs[2], 16x3x256x480) and of the uest train step on one GPU."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import models, training, uest
from tests.synth import synth_state_dict

dev = 'cuda'
a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
x = torch.randn(16, 3, 256, 480, device=dev)
nets = []
for i, (C, ds) in enumerate([(13, 'camvid'), (20, 'city'), (5, 'forest')]):
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=C, dataset=ds, fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(m.state_dict(), i))
    nets.append(m)
p = uest.PseudoLabelPass(nets, ['camvid', 'cityscapes', 'forest'], merge_label_policy='all', device=dev, use_graph=True)
labels = p(x).clone()
for _ in range(5):
    p(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    p(x)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
print('3-source label pass: %.2f ms/batch16 -> %.0f img/s' % (t * 1e3, 16 / t))
for depth in (2, 3):
    plp = uest.PipelinedLabelPass(lambda: uest.PseudoLabelPass(nets, ['camvid', 'cityscapes', 'forest'], merge_label_policy='all',
                                                               device=dev, use_graph=True), depth=depth, device=dev)
    for _ in range(2 * depth + 2):
        plp(x)
    list(plp.flush())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(24):
        plp(x)
    list(plp.flush())
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 24
    print('3-source label pass, %d batches in flight: %.2f ms/batch16 -> %.0f img/s' % (depth, t * 1e3, 16 / t))
    del plp

tgt = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
tgt.load_state_dict(synth_state_dict(tgt.state_dict(), 9))
tgt = tgt.to(dev).eval()
y = labels.to(torch.int64)
cw = torch.ones(5)
loss, opt = training.train_step(tgt, x, y, cw, None, ignore_idx=4)
for _ in range(2):
    training.train_step(tgt, x, y, cw, opt, ignore_idx=4)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    loss, opt = training.train_step(tgt, x, y, cw, opt, ignore_idx=4)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
print('train step (fwd + UW-loss + bwd + Adam), bs16 256x480: %.1f ms -> %.1f img/s, loss %.4f' % (t * 1e3, 16 / t, float(loss)))

tgt2 = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
tgt2.load_state_dict(synth_state_dict(tgt2.state_dict(), 9))
tgt2 = tgt2.to(dev).eval()
gs = training.GraphedTrainStep(tgt2, x, y, cw, ignore_idx=4)
for _ in range(2):
    gs(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    loss = gs(x, y)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
print('train step, hipGraph replay: %.1f ms -> %.1f img/s, loss %.4f' % (t * 1e3, 16 / t, float(loss)))
