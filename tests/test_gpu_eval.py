"""The evaluation step on the GPU (mspl_amd.evaluation: EvalPass / val_seg_ue) against the reference's own val_seg_ue and test()
bodies (tests/golden/eval.npz, produced by AST-extracting val_seg_ue from utilities/train_eval_seg.py and running it with the
reference's model, SegmentationLoss and MIOU on a seeded loader) and against the CPU oracle at awkward shapes."""
import argparse

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import labels as olab
from oracle import net as onet
from tests.cases import EVAL_CASES
from tests.synth import synth_eval_batches, synth_state_dict

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _model(C, ds, seed):
    from mspl_amd import models
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=C, dataset=ds, fix_pyr_plane_proj=True)
    sd = synth_state_dict(m.state_dict(), seed)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


@pytest.mark.parametrize('name', sorted(EVAL_CASES))
@pytest.mark.parametrize('use_graph', [False, True])
def test_eval_step_vs_reference_golden(name, use_graph, golden):
    from mspl_amd import evaluation as ev
    g = golden('eval')
    C, ds, shape, nb, sd_seed, in_seed, ign, cw_seed, with_void = EVAL_CASES[name]
    m, _ = _model(C, ds, sd_seed)
    cw = torch.from_numpy(g[name + '.cw'])
    loader = synth_eval_batches(EVAL_CASES[name])

    class Crit:
        loss_type, class_wts, ignore_idx = 'ce', cw, ign
    iou, loss = ev.val_seg_ue(m, loader, criterion=Crit(), num_classes=C, device=DEV, use_graph=use_graph)
    # areas are integer counts: bit-exact unless a pixel's top-2 margin is at rounding level; the golden iou pins them to 1e-6
    np.testing.assert_allclose(iou, g[name + '.iou'], rtol=0, atol=2e-6)
    assert abs(loss - float(g[name + '.loss'])) < 1e-5 * max(1.0, abs(float(g[name + '.loss'])))
    # test() of the uest script: main head alone
    ep = ev.EvalPass(m, C, class_weights=cw, ignore_idx=ign, aux_weight=0.0, device=DEV, use_graph=use_graph)
    for x, y in loader:
        ep(x, y)
    iou0, loss0 = ep.result()
    np.testing.assert_allclose(iou0, g[name + '.test_iou'], rtol=0, atol=2e-6)
    assert abs(loss0 - float(g[name + '.test_loss'])) < 1e-5 * max(1.0, abs(float(g[name + '.test_loss'])))
    # MIOU only (criterion=None): the loss slot is 0 like the reference's return
    iou1, zero = ev.val_seg_ue(m, loader, criterion=None, num_classes=C, device=DEV, use_graph=False)
    assert zero == 0
    np.testing.assert_allclose(iou1, g[name + '.iou'], rtol=0, atol=2e-6)


@pytest.mark.parametrize('cfg', [(2, 5, 20, 33, 10, 17, 40, 66, 0.5, 4), (1, 13, 36, 60, 18, 30, 72, 120, 0.5, 255),
                                 (3, 20, 9, 300, 5, 150, 17, 600, 0.0, 255), (2, 7, 16, 16, 0, 0, 32, 32, 0.5, 2)])
def test_eval_epilogue_kernel_vs_torch(cfg):
    """mspl_eval_epilogue_fwd on bare heads against the definition (ATen up-sampling, cross entropy sums, the oracle's MIOU areas):
    odd sizes, more columns than one workgroup, single-head form, ignore index inside the class range; accumulation over two calls."""
    from mspl_amd import evaluation as ev
    N, C, Hm, Wm, Ha, Wa, H, W, aw, ign = cfg
    g = torch.Generator().manual_seed(5)
    main = torch.randn(N, C, Hm, Wm, generator=g) * 2
    aux = torch.randn(N, C, Ha, Wa, generator=g) * 2 if Ha else None
    tgt = torch.randint(0, C, (N, H, W), generator=g)
    tgt[torch.rand(N, H, W, generator=g) < 0.1] = 255 if ign == 255 else ign
    cw = torch.rand(C, generator=g) + 0.5
    mu = F.interpolate(main, (H, W), mode='bilinear', align_corners=True)
    o = mu + aw * F.interpolate(aux, (H, W), mode='bilinear', align_corners=True) if (aux is not None and aw != 0) else mu
    valid = (tgt != ign) & (tgt >= 0) & (tgt < C)
    nll = F.cross_entropy(o, torch.where(valid, tgt, torch.zeros_like(tgt)), reduction='none')
    wt = cw[torch.where(valid, tgt, torch.zeros_like(tgt))] * valid
    K = C - 1
    inter, union = olab.miou_areas(o, tgt, K)
    sums = torch.zeros(2, dtype=torch.float64, device=DEV)
    areas = torch.zeros((3, K), dtype=torch.int64, device=DEV)
    labels = torch.empty((N, H, W), dtype=torch.uint8, device=DEV)
    for _ in range(2):                                   # accumulates
        ev.eval_epilogue(main.to(DEV), None if aux is None else aux.to(DEV), tgt.to(DEV), cw.to(DEV), (H, W), aw, ign, K, sums, areas,
                         labels)
    s = sums.cpu().numpy() / 2
    assert abs(s[0] - float((nll * wt).double().sum())) < 2e-5 * max(1.0, abs(s[0]))
    assert abs(s[1] - float(wt.double().sum())) < 1e-6 * max(1.0, s[1])
    a = areas.cpu().numpy().astype(np.float64) / 2
    srt = torch.sort(o, 1, descending=True)[0]
    ties = int(((srt[:, 0] - srt[:, 1]) <= 1e-5).sum())  # pixels whose argmax two fp32 evaluations may order differently
    assert np.abs(a[0] - inter).max() <= ties and np.abs(a[1] + a[2] - a[0] + 1e-6 - union).max() <= 2 * ties + 1e-3
    same = (labels.cpu() == o.argmax(1).to(torch.uint8))
    assert int((~same).sum()) <= ties


def test_eval_pass_graph_equals_eager_and_reset():
    from mspl_amd import evaluation as ev
    C = 5
    m, _ = _model(C, 'greenhouse', 71)
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn(2, 3, 64, 96, generator=g).to(DEV) for _ in range(3)]
    ys = [torch.randint(0, C, (2, 64, 96), generator=g).to(DEV) for _ in range(3)]
    a, b = ev.EvalPass(m, C, ignore_idx=4, device=DEV, use_graph=False), ev.EvalPass(m, C, ignore_idx=4, device=DEV, use_graph=True)
    for x, y in zip(xs, ys):
        a(x, y)
        b(x, y)
    assert torch.equal(a.areas, b.areas) and a.batches == b.batches == 3
    assert torch.allclose(a.acc[:2], b.acc[:2], rtol=1e-12, atol=0)
    b.reset()
    assert int(b.areas.sum()) == 0 and b.batches == 0 and float(b.acc.sum()) == 0.0
    b(xs[0], ys[0])
    c = ev.EvalPass(m, C, ignore_idx=4, device=DEV)
    c(xs[0], ys[0])
    assert torch.equal(b.areas, c.areas)


@pytest.mark.parametrize('depth,group', [(3, 1), (3, 2), (2, 3)])
def test_pipelined_eval_pass_equals_one_lane(depth, group):
    """Three evaluation steps in flight (PipelinedEvalPass, what val_seg_ue runs) give the sums of one lane: areas exactly, the loss sums
    to float64 rounding; batches of two shapes (a short last batch), reset, and the tensors handed in may be reused at once."""
    from mspl_amd import evaluation as ev
    C = 5
    m, _ = _model(C, 'greenhouse', 72)
    g = torch.Generator().manual_seed(10)
    shapes = [(4, 64, 96)] * 7 + [(2, 64, 96)]       # (with group > 1: a partly filled lane and a ragged batch at the end)
    xs = [torch.randn(n, 3, h, w, generator=g) for n, h, w in shapes]
    ys = [torch.randint(0, C, (n, h, w), generator=g) for n, h, w in shapes]
    one = ev.EvalPass(m, C, class_weights=torch.rand(C, generator=g) + 0.5, ignore_idx=4, device=DEV, use_graph=True)
    three = ev.PipelinedEvalPass(m, C, depth=depth, group=group, class_weights=one.cw, ignore_idx=4, device=DEV)
    buf_x, buf_y = torch.empty(4, 3, 64, 96, device=DEV), torch.empty(4, 64, 96, dtype=torch.int64, device=DEV)
    for rep in range(2):
        for x, y in zip(xs, ys):
            one(x.to(DEV), y.to(DEV))
            n = x.shape[0]
            buf_x[:n].copy_(x)                      # the same device buffers every batch: the lane must have taken its copy before
            buf_y[:n].copy_(y)                      # the next batch overwrites them
            three(buf_x[:n], buf_y[:n])
        a, b = one.sums().cpu().numpy(), three.sums().cpu().numpy()
        assert three.batches == one.batches == len(xs)
        assert np.array_equal(a[:3 * (C - 1)], b[:3 * (C - 1)]) and a[-1] == b[-1] and a[-2] == b[-2]
        assert abs(a[-3] - b[-3]) <= 1e-12 * abs(a[-3])
        i1, l1 = one.result(reduce=False)
        i3, l3 = three.result(reduce=False)
        assert np.array_equal(i1, i3) and abs(l1 - l3) <= 1e-12 * abs(l1)
        one.reset()
        three.reset()
        assert three.batches == 0 and float(three.sums().abs().sum()) == 0.0
