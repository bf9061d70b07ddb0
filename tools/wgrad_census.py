"""The 1x1 weight-gradient problems of one uest train step (16 x 3 x 256x480) as the queue sees them: shapes, operand bytes, flops, and
which launch (flush of <= 64 problems, runs of <= 16) they go out in."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import models, training, autograd as ag
from tests.synth import synth_state_dict
a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
m = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
m.load_state_dict(synth_state_dict(m.state_dict(), 9))
m = m.cuda().eval()
x = torch.randn(16, 3, 256, 480).cuda(); y = torch.randint(0, 5, (16, 256, 480)).cuda(); cw = torch.ones(5)
l, opt = training.train_step(m, x, y, cw, None, ignore_idx=4)
log = []
orig_add, orig_flush = ag.WGRADS.add, ag.WGRADS.flush
def add(gy, x_, N, Cin, Cout, groups, HW, sink, rowscale=None):
    log.append(('p', N, Cin, Cout, groups, HW, rowscale is not None))
    return orig_add(gy, x_, N, Cin, Cout, groups, HW, sink, rowscale)
def flush():
    log.append(('flush', len(ag.WGRADS.items)))
    return orig_flush()
ag.WGRADS.add, ag.WGRADS.flush = add, flush
training.train_step(m, x, y, cw, opt, ignore_idx=4)
torch.cuda.synchronize()
tot_b = tot_f = 0
for e in log:
    if e[0] == 'flush':
        print('--- flush of', e[1]); continue
    _, N, Cin, Cout, g, HW, rs = e
    b = 4.0 * N * HW * (Cin + Cout); f = 2.0 * N * HW * Cin * Cout / g
    tot_b += b; tot_f += f
    print('N %2d Cin %4d Cout %4d groups %2d HW %6d  %7.1f MB  %6.2f GFLOP %s' % (N, Cin, Cout, g, HW, b / 1e6, f / 1e9, 'rowscale' if rs else ''))
print('total %.1f MB operands, %.1f GFLOP' % (tot_b / 1e6, tot_f / 1e9))
