"""BASELINE.json's full sizes (bs=16, 288x480 / 256x480) through size-independent properties, plus spot checks of single
images against the CPU oracle.  Needs a real MI355X: run with `-m gpu`.

Properties used: an image's labels / KL map do not depend on its batch-mates (batch 16 == batch 1, bit for bit: every
kernel works per image plane or per pixel); hipGraph replay == eager launches; the class histogram counts every pixel
once; merging is a pure function of the per-source maps (bit-exact against the oracle's merge of the SAME maps); merging S
copies of a map returns the map wherever its class is a valid greenhouse class; a graphed train step equals an eager one.
"""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from oracle import labels as olab
from oracle import net as onet
from tests.conftest import GOLDEN
from tests.synth import synth_input, synth_labels, synth_state_dict

pytestmark = pytest.mark.gpu
KEYS = json.load(open(os.path.join(GOLDEN, 'state_dict_keys.json')))
DEV = 'cuda'


def _net(C, ds, seed):
    from mspl_amd import models as M
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    m = M.ESPDNetwithUncertaintyEstimation(a, classes=C, dataset=ds, fix_pyr_plane_proj=True)
    sd = synth_state_dict(KEYS['espdnetue_s2.0_c%d' % C], seed)
    m.load_state_dict(sd)
    return m, sd


def test_config2_self_label_pass_bs16_288x480():
    """BASELINE configs[1]: ESPDNet-UE s=2.0 C=13 single-source label pass, 16x3x288x480."""
    from mspl_amd import uest
    m, sd = _net(13, 'camvid', 0)
    x = synth_input((16, 3, 288, 480), 77)
    eager = uest.SelfLabelPass(m, classes=13, device=DEV, use_graph=False)
    lab, kld = eager(x.to(DEV))
    lab, kld = lab.clone(), kld.clone()
    assert lab.shape == (16, 288, 480) and lab.dtype == torch.uint8 and kld.shape == (16, 288, 480)
    assert int(eager.hist.sum()) == 16 * 288 * 480                       # every pixel counted once
    np.testing.assert_array_equal(eager.hist.cpu().numpy(), np.bincount(lab.cpu().numpy().ravel(), minlength=13))
    assert torch.isfinite(kld).all() and float(kld.min()) > -1e-5        # a KL divergence
    # hipGraph replay == eager, twice (static buffers are reused)
    graphed = uest.SelfLabelPass(m, classes=13, device=DEV, use_graph=True)
    for _ in range(2):
        lg, kg = graphed(x.to(DEV))
        assert torch.equal(lg, lab) and torch.equal(kg, kld)
    # batch independence: image i alone gives the same bits
    solo = uest.SelfLabelPass(m, classes=13, device=DEV, use_graph=False)
    for i in (0, 7, 15):
        li, ki = solo(x[i:i + 1].to(DEV))
        assert torch.equal(li[0], lab[i]) and torch.equal(ki[0], kld[i])
    # one image against the CPU oracle at full size
    with torch.no_grad():
        main, aux = onet.espdnet_ue_forward(sd, x[3:4])
        prob, kref = olab.get_output(main, aux)
    ref = olab.argmax_labels(prob)[0]
    top2 = np.sort(prob.numpy()[0], axis=0)[-2:]
    clear = (top2[1] - top2[0]) > 1e-4                                   # away from fp32 ties of the reference itself
    got = lab[3].cpu().numpy()
    assert (got == ref)[clear].all() and clear.mean() > 0.99
    np.testing.assert_allclose(kld[3].cpu().numpy(), kref.numpy()[0], rtol=0, atol=2e-4)


def test_config3_three_source_pass_bs16_256x480():
    """BASELINE configs[2]: CamVid(13) + Cityscapes(20) + Forest(5) -> LUT -> merge('all') -> histogram, 16x3x256x480."""
    from mspl_amd import ops, uest
    specs = [(13, 'camvid', 'camvid'), (20, 'city', 'cityscapes'), (5, 'forest', 'forest')]
    nets = [_net(C, ds, 60 + i)[0] for i, (C, ds, _) in enumerate(specs)]
    x = synth_input((16, 3, 256, 480), 5)
    p = uest.PseudoLabelPass(nets, [s[2] for s in specs], merge_label_policy='all', device=DEV, use_graph=True)
    merged = p(x.to(DEV)).clone()
    maps = [t.clone() for t in p.source_maps(x.to(DEV))]
    np_maps = np.stack([t.cpu().numpy() for t in maps])
    ref = olab.merge_outputs(np_maps.reshape(3, -1, 480), 5, 'all').reshape(16, 256, 480)     # integer stage: bit-exact
    np.testing.assert_array_equal(merged.cpu().numpy(), ref.astype(np.uint8))
    p.reset()
    again = p(x.to(DEV))
    assert torch.equal(again, merged)
    np.testing.assert_array_equal(p.hist.cpu().numpy(), np.bincount(ref.ravel(), minlength=5)[:5])
    assert int(p.hist.sum()) == 16 * 256 * 480
    # merging S copies of one map is the identity on valid classes (and 4 elsewhere)
    same = ops.merge_labels([maps[0]] * 3, 5, 3, 4).cpu().numpy()
    m0 = maps[0].cpu().numpy()
    np.testing.assert_array_equal(same, np.where(m0 < 5, m0, 4).astype(np.uint8))
    # batch independence of the whole multi-source pass
    p1 = uest.PseudoLabelPass(nets, [s[2] for s in specs], merge_label_policy='all', device=DEV, use_graph=False)
    assert torch.equal(p1(x[9:10].to(DEV))[0], merged[9])


def test_config3_train_step_bs16_256x480():
    """The uest train step at the benchmark size: finite loss and gradients, graphed == eager, loss goes down."""
    from mspl_amd import training
    x = synth_input((16, 3, 256, 480), 8).to(DEV)
    y = synth_labels((16, 256, 480), 5, 8).to(DEV)
    cw = torch.ones(5)
    nets = [_net(5, 'greenhouse', 3)[0].to(DEV).eval() for _ in range(2)]
    l0, opt = training.train_step(nets[0], x, y, cw, None, ignore_idx=4)
    assert all(torch.isfinite(p.grad).all() for p in opt.params)
    eager = [float(l0)]
    for _ in range(3):
        l, opt = training.train_step(nets[0], x, y, cw, opt, ignore_idx=4)
        eager.append(float(l))
    gs = training.GraphedTrainStep(nets[1], x, y, cw, ignore_idx=4)
    graphed = [float(gs(x, y)) for _ in range(2)]
    assert np.isfinite(eager).all() and eager[-1] < eager[0]
    np.testing.assert_allclose(graphed, eager[2:], rtol=5e-4)
    # bench.py's configuration: the batch as four concurrent micro-batch graphs -- same losses, same weights
    net4 = _net(5, 'greenhouse', 3)[0].to(DEV).eval()
    g4 = training.GraphedTrainStep(net4, x, y, cw, ignore_idx=4, lanes=4)
    assert g4.lanes == 4
    lanes = [float(g4(x, y)) for _ in range(2)]
    np.testing.assert_allclose(lanes, eager[2:], rtol=5e-4)
    worst = max(float((p - q).abs().max()) for p, q in zip(net4.state_dict().values(), nets[1].state_dict().values()))
    assert worst < 1e-4, worst


@pytest.mark.parametrize('depth,group', [(2, 1), (3, 1), (3, 2), (2, 3)])
def test_pipelined_label_pass_equals_single_lane(depth, group):
    """`depth` label passes in flight (PipelinedLabelPass, hipGraph lanes on their own streams; 3 = bench.py's default): the same
    label maps, uncertainty maps and class histogram as one pass at a time, batch by batch, at the BASELINE shape.  group > 1:
    that many consecutive batches per launch (bench.py: 2); 7 batches leave a partly filled lane for flush().
    Bit equality is a CONTRACT here, not an accident of summation order: the reference pass is captured with the same launch
    flags as the lanes (MSPL_LAUNCH_THROUGHPUT: K1 and K2 as two launches), so both sides run the same kernels on the same inputs;
    images are independent, so neither the lane nor the batches-per-launch may change a bit.  (The fused K1+K2 form a lone pass
    uses by default may differ from the two-launch form in the last bits: test_eesp_proj_dw_hff covers that with a tolerance.)"""
    import argparse
    from mspl_amd import models, ops, uest
    from tests.synth import synth_state_dict
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=13, dataset='camvid', fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(m.state_dict(), 3))
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(77)
    batches = [torch.randn((16, 3, 288, 480), generator=g).cuda() for _ in range(7)]
    ref = uest.SelfLabelPass(m, classes=13, use_graph=True)
    want = []
    with ops.launch_flags(throughput=True):
        for b in batches:
            lab, kld = ref(b)
            want.append((lab.clone(), kld.clone()))
    plp = uest.PipelinedLabelPass(lambda: uest.SelfLabelPass(m, classes=13, use_graph=True), depth=depth, group=group)
    got = []
    for b in batches:
        out = plp(b)
        if out is not None:
            got.append((out[0].clone(), out[1].clone()))
    got += [(o[0].clone(), o[1].clone()) for o in plp.flush()]
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for (l1, k1), (l0, k0) in zip(got, want):
        assert torch.equal(l1, l0) and torch.equal(k1, k0)
    assert torch.equal(plp.hist, ref.hist) and int(plp.hist.sum()) == 7 * 16 * 288 * 480


def test_pipelined_histogram_without_a_writer():
    """on_lane=True hands results back without making the current stream wait for the lane; nothing else (no LabelWriter, no
    synchronize) orders the lanes before the histogram is read: PipelinedLabelPass.hist must join the lanes itself."""
    import argparse
    from mspl_amd import models, uest
    from tests.synth import synth_state_dict
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(m.state_dict(), 5))
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(78)
    batches = [torch.randn((16, 3, 256, 480), generator=g).cuda() for _ in range(6)]
    plp = uest.PipelinedLabelPass(lambda: uest.SelfLabelPass(m, classes=5, use_graph=True, with_kld=False), depth=3, group=2)
    for rep in range(2):
        plp.reset()
        torch.cuda.synchronize()
        for b in batches:
            plp(b, on_lane=True)
        for _ in plp.flush(on_lane=True):
            pass
        h = plp.hist                       # no synchronize, no writer in between
        assert int(h.sum()) == 6 * 16 * 256 * 480


def test_graphed_pass_follows_parameter_updates():
    """A captured pass bakes in pointers to the folded-BN / packed-weight caches and to the parameter storages.  The uest loop
    alternates label passes and train rounds on the SAME model: after a train step (FlatAdam re-points .data, the Adam kernel
    writes through raw pointers) the graphed pass must re-capture and equal a fresh eager pass, not replay stale weights."""
    from mspl_amd import training, uest
    m = _net(5, 'greenhouse', 4)[0].to(DEV).eval()
    x = synth_input((4, 3, 128, 160), 21).to(DEV)
    y = synth_labels((4, 128, 160), 5, 21).to(DEV)
    graphed = uest.SelfLabelPass(m, classes=5, device=DEV, use_graph=True)
    lab0, kld0 = [t.clone() for t in graphed(x)]
    g0 = graphed._graphs[tuple(x.shape)].graph
    assert graphed(x) is not None and graphed._graphs[tuple(x.shape)].graph is g0          # unchanged parameters: same graph
    opt = None
    for _ in range(3):
        _, opt = training.train_step(m, x, y, torch.ones(5), opt, ignore_idx=4, lr=1e-2)
    lab1, kld1 = [t.clone() for t in graphed(x)]
    assert graphed._graphs[tuple(x.shape)].graph is not g0                                 # re-captured
    fresh = uest.SelfLabelPass(m, classes=5, device=DEV, use_graph=False)
    lab2, kld2 = fresh(x)
    assert torch.equal(lab1, lab2) and torch.equal(kld1, kld2)
    assert not torch.equal(kld1, kld0)                                                     # the step really moved the weights
    # in-place edits through torch (load_state_dict -> copy_) are seen through the version counters
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd['bu_dec_l4.merge_layer.3.bias'] += 1.5
    m.load_state_dict(sd)
    lab3, _ = graphed(x)
    assert torch.equal(lab3, uest.SelfLabelPass(m, classes=5, device=DEV, use_graph=False)(x)[0])


def test_graph_outlives_its_pass_object():
    """Round-1 fault on record: a graph replayed after its SelfLabelPass had been collected read a freed
    input buffer.  Now the graph object itself references every buffer it was captured on: drop the pass, capture another lane
    (torch.cuda.graph runs gc.collect() + empty_cache()), replay the first graph -- same bits as before."""
    import gc
    from mspl_amd import uest
    m = _net(13, 'camvid', 6)[0].to(DEV).eval()
    x = synth_input((4, 3, 128, 160), 31).to(DEV)
    pa = uest.SelfLabelPass(m, classes=13, device=DEV, use_graph=True)
    want = [t.clone() for t in pa(x)]
    cap = pa._graphs[tuple(x.shape)]
    graph, out = cap.graph, cap.static_out
    hist_ref = pa.hist
    del pa, cap
    gc.collect()
    torch.cuda.empty_cache()
    pb = uest.SelfLabelPass(m, classes=13, device=DEV, use_graph=True)                      # capture lane B in the freed space
    junk = [torch.full((4, 3, 128, 160), float(i), device=DEV) for i in range(8)]           # and recycle whatever is left
    pb(x * 0.5)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], want[0]) and torch.equal(out[1], want[1])
    assert int(hist_ref.sum()) == 2 * 4 * 128 * 160
    del junk
    # the handed-out input view carries the graph along
    pc = uest.SelfLabelPass(m, classes=13, device=DEV, use_graph=True)
    pc(x)
    view = pc.static_input(x.shape)
    assert view is not None and view._mspl_graph is pc._graphs[tuple(x.shape)].graph


def test_aspp_dense_conv_full_size_properties():
    """BASELINE configs[4] at the bench shape 16 x 2048 x 32 x 64 (no oracle run at this size: 29 GMAC per image).  Properties:
    image i alone gives the same bits (every workgroup tile lies inside one image); writing into a channel slice of a wider
    destination leaves the guard bands untouched and gives the same bits as a dense destination."""
    from mspl_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn((16, 2048, 32, 64), generator=g).to(DEV)
    w = (torch.randn((256, 2048, 3, 3), generator=g) * 0.01).to(DEV)
    wp = ops.pack_dense_weight(w)
    scale = (torch.rand(256, generator=g) + 0.5).to(DEV)
    shift = torch.randn(256, generator=g).to(DEV)
    for dil in (6, 18):
        full = ops.dense_conv(x, wp, 3, dil, ops.Epi(scale, shift, torch.zeros(256, device=DEV)))
        assert torch.isfinite(full).all()
        for i in (0, 9, 15):
            one = ops.dense_conv(x[i:i + 1].contiguous(), wp, 3, dil, ops.Epi(scale, shift, torch.zeros(256, device=DEV)))
            assert torch.equal(one[0], full[i])
    wide = torch.full((16, 256 + 64, 32, 64), -7.0, device=DEV)
    sc = torch.cat([torch.ones(32, device=DEV), scale, torch.ones(32, device=DEV)])
    sh = torch.cat([torch.zeros(32, device=DEV), shift, torch.zeros(32, device=DEV)])
    ops.dense_conv(x, wp, 3, 18, ops.Epi(sc, sh, torch.zeros(320, device=DEV)), out=(wide, 32))
    assert torch.equal(wide[:, 32:288], full)
    assert (wide[:, :32] == -7.0).all() and (wide[:, 288:] == -7.0).all()
    # a corner pixel of one image against a direct fp64 evaluation of the definition (dilation 18: 4 of 9 taps inside)
    ref = 0.0
    xi = x[2].double()
    for ky in range(3):
        for kx in range(3):
            yy, xx = 0 + (ky - 1) * 18, 0 + (kx - 1) * 18
            if 0 <= yy < 32 and 0 <= xx < 64:
                ref = ref + (w[:, :, ky, kx].double() @ xi[:, yy, xx])
    ref = torch.relu(ref * scale.double() + shift.double())
    np.testing.assert_allclose(full[2, :, 0, 0].cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-4, atol=2e-4)


def test_pipelined_static_inputs_after_partial_launch():
    """PipelinedLabelPass.next_lane after an odd number of batches + flush() (a partly filled lane was labelled by a shorter launch):
    the index must name the slot the NEXT batch is staged into -- lane (launches so far) % depth, slot 0 -- not a position derived from
    the number of calls.  Writing a batch into static_inputs()[next_lane] and submitting that view must neither copy the batch (the
    submit sees its own buffer) nor label another lane's data (round 2's index pointed at a slot whose launch had just been issued)."""
    import argparse
    from unittest import mock
    from mspl_amd import models, uest
    from tests.synth import synth_state_dict
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(m.state_dict(), 5))
    m = m.cuda().eval()
    shape = (4, 3, 128, 160)
    g = torch.Generator().manual_seed(31)
    batches = [torch.randn(shape, generator=g).cuda() for _ in range(12)]
    ref = uest.SelfLabelPass(m, classes=5, use_graph=False)
    want = [ref(b)[0].clone() for b in batches]
    plp = uest.PipelinedLabelPass(lambda: uest.SelfLabelPass(m, classes=5, use_graph=True), depth=3, group=2)
    for b in batches[:6]:                    # every lane captures its full-group graph
        plp(b)
    list(plp.flush())
    for b in batches[:5]:                    # odd: the third launch is partly filled
        plp(b)
    list(plp.flush())
    xs = plp.static_inputs(shape)
    assert all(x is not None for x in xs) and len(xs) == 6
    big_copies = []
    real_copy = torch.Tensor.copy_

    def counting_copy(self, src, *a_, **kw):
        if self.numel() >= batches[0].numel():
            big_copies.append(tuple(self.shape))
        return real_copy(self, src, *a_, **kw)
    got, slots = [], []
    for i, b in enumerate(batches[5:12]):
        k = plp.next_lane
        slots.append(k)
        xs[k].copy_(b)                        # the caller fills the slot it was told about (on the current stream)
        with mock.patch.object(torch.Tensor, 'copy_', counting_copy):
            out = plp(xs[k])
        if out is not None:
            got.append(out[0].clone())
    got += [o[0].clone() for o in plp.flush()]
    torch.cuda.synchronize()
    assert not big_copies, big_copies        # every submit found the batch already in its staging slot
    # after 3 + 3 launches the next lane is 0: slots 0,1 (lane 0), 2,3 (lane 1), 4,5 (lane 2), then lane 0 again
    assert slots == [0, 1, 2, 3, 4, 5, 0], slots
    assert len(got) == 7
    for l1, l0 in zip(got, want[5:12]):
        assert torch.equal(l1, l0)
