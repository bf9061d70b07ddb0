"""Where do the gradients of the sink path and of autograd's own accumulation part ways?  python tools/diag_sinks.py STEPS
prints, per parameter, the largest difference after STEPS train steps at 64x96 for repeated runs of both settings."""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import models, training, autograd as ag
from tests.synth import synth_input, synth_labels, synth_state_dict
KEYS = json.load(open('tests/golden/state_dict_keys.json'))
a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
hw = (64, 96)
x = synth_input((2, 3) + hw, 18).cuda(); y = synth_labels((2,) + hw, 5, 18).cuda(); cw = torch.ones(5)
res = {}
for flag in ('1', '0', 'q0', '0b', '1b'):
    os.environ['MSPL_GRAD_SINKS'] = '0' if flag[0] == '0' else '1'
    ag.WGRADS.enabled = flag != 'q0'
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(KEYS['espdnetue_s2.0_c5'], 4)); m = m.cuda().eval()
    opt = None
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
        l, opt = training.train_step(m, x, y, cw, opt, ignore_idx=4)
    res[flag] = (float(l), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
for other in ('1', 'q0', '0b', '1b'):
    print('--- sinks(%s) vs autograd accumulation; losses' % other, res[other][0], res['0'][0])
    worst = []
    for n, g0 in res['0'][1].items():
        g1 = res[other][1][n]
        d = float((g1 - g0).abs().max()); s = float(g0.abs().max()) + 1e-12
        worst.append((d / s, d, s, n))
    worst.sort(reverse=True)
    for w in worst[:8]: print('  rel %.2e abs %.2e max %.2e %s' % w)
