"""CPU-side checks of the product's host logic: C-ABI symbols, state-dict parity with the reference,
checkpoint-loader quirks, drop-in aliases, and loud failure without a GPU.  No kernel is launched."""
import argparse
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import mspl_amd
from mspl_amd import _native, layers, models, uest
from tests.cases import LAYER_CASES
from tests.conftest import GOLDEN, ROOT
from tests.synth import synth_state_dict

KEYS = json.load(open(os.path.join(GOLDEN, 'state_dict_keys.json')))
LAYER_KEYS = json.load(open(os.path.join(GOLDEN, 'layer_keys.json')))


def _args(s):
    return argparse.Namespace(s=s, channels=3, num_classes=1000)


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, 'include', 'mspl_hip.h')).read()
    declared = set(re.findall(r'\b(mspl_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'mspl_status'}
    assert declared, 'no declarations parsed'
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), 'include/mspl_hip.h declares %s but the library does not export it' % name
    # and the Python binding table covers every compute entry point
    assert set(_native.SIGNATURES) == declared - {'mspl_version', 'mspl_last_error'}
    assert 'gfx950' in _native.version()


@pytest.mark.parametrize('cfg', sorted(KEYS))
def test_model_state_dict_matches_reference(cfg):
    kind, s, c = re.match(r'(\w+)_s([\d.]+)_c(\d+)', cfg).groups()
    s, c = float(s), int(c)
    ds = {13: 'camvid', 5: 'greenhouse', 20: 'city'}[c]
    if kind == 'espdnetue':
        m = models.ESPDNetwithUncertaintyEstimation(_args(s), classes=c, dataset=ds, fix_pyr_plane_proj=True)
    else:
        m = models.ESPNetv2Segmentation(_args(s), classes=c, dataset=ds)
    got = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert got == KEYS[cfg]


@pytest.mark.parametrize('name', sorted(LAYER_CASES))
def test_layer_state_dict_matches_reference(name):
    kind, kw, _, _ = LAYER_CASES[name]
    cls = {'eesp': layers.EESP, 'down': layers.DownSampler, 'pyr': layers.EfficientPyrPool,
           'pw': layers.EfficientPWConv}[kind]
    got = {k: list(v.shape) for k, v in cls(**kw).state_dict().items()}
    assert got == LAYER_KEYS[name]


def test_eesp_dilations():
    assert layers.eesp_dilations(13) == [1, 2, 3, 4]
    assert layers.eesp_dilations(9) == [1, 2, 3, 4]
    assert layers.eesp_dilations(7) == [1, 1, 2, 3]
    assert layers.eesp_dilations(5) == [1, 1, 1, 2]


def test_param_groups_and_unused_parameters():
    m = models.ESPDNetwithUncertaintyEstimation(_args(2.0), classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
    assert sum(p.numel() for p in m.parameters()) == 2234230
    assert len(list(m.parameters())) == 570
    base = sum(p.numel() for p in m.get_basenet_params())
    seg = sum(p.numel() for p in m.get_segment_params())
    depth = sum(p.numel() for p in m.get_depth_encoder_params())
    assert base == sum(p.numel() for p in m.base_net.parameters())
    assert depth == sum(p.numel() for p in m.depth_base_net.parameters())
    assert seg > 0


def test_lossy_checkpoint_loader(tmp_path):
    """espdnetue_seg2 refills depth_base_net from the file's base_net.* entries."""
    a = _args(2.0)
    src = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
    with torch.no_grad():
        for p in src.parameters():
            p.add_(torch.randn_like(p) * 0.01)
    path = str(tmp_path / 'ckpt.pth')
    torch.save(src.state_dict(), path)
    a2 = argparse.Namespace(s=2.0, channels=3, num_classes=1000, classes=5, dataset='greenhouse', weights=path,
                            trainable_fusion=True, dense_fuse=False)
    m = models.espdnetue_seg2(a2, load_entire_weights=True, fix_pyr_plane_proj=True)
    sd, ref = m.state_dict(), src.state_dict()
    assert torch.equal(sd['base_net.level3.1.proj_1x1.conv.weight'], ref['base_net.level3.1.proj_1x1.conv.weight'])
    # the depth slot now holds the RGB encoder's tensor, not what was saved in that slot
    assert torch.equal(sd['depth_base_net.level3.1.proj_1x1.conv.weight'], ref['base_net.level3.1.proj_1x1.conv.weight'])
    assert not torch.equal(sd['depth_base_net.level3.1.proj_1x1.conv.weight'],
                           ref['depth_base_net.level3.1.proj_1x1.conv.weight'])
    assert torch.allclose(sd['depth_base_net.level1.conv.weight'], ref['base_net.level1.conv.weight'].mean(1, keepdim=True))
    assert torch.equal(sd['aux_decoder.stages.0.weight'], ref['aux_decoder.stages.0.weight'])


def test_thresh_rule():
    assert uest.resolve_thresh(3, None) == 2
    assert uest.resolve_thresh(3, 'half') == 2
    assert uest.resolve_thresh(3, 'all') == 3
    assert uest.resolve_thresh(3, 'bogus') == 2
    assert uest.resolve_thresh(3, 1) == 1
    assert uest.resolve_thresh(3, 7) == 2
    assert uest.resolve_thresh(2, None) == 2
    assert uest.resolve_thresh(1, 'all') == 1


def test_luts_match_reference_tables():
    g = np.load(os.path.join(GOLDEN, 'labels.npz'))
    np.testing.assert_array_equal(uest.id_camvid_to_greenhouse, g['lut_id_camvid_to_greenhouse'])
    np.testing.assert_array_equal(uest.id_cityscapes_to_greenhouse, g['lut_id_cityscapes_to_greenhouse'])
    np.testing.assert_array_equal(uest.id_forest_to_greenhouse, g['lut_id_forest_to_greenhouse'])


def test_class_weights_rule():
    w = uest.class_weights_from_histogram([10, 20, 30, 40, 0])
    assert w[0] == 0.0 and abs(w[1] - 5.0) < 1e-6 and w[4] == pytest.approx(1e10)
    assert np.array_equal(uest.class_weights_from_histogram([1, 2, 3], 'other'), np.ones(3))


def test_dropin_aliases():
    import sys
    for n in [k for k in sys.modules if k.split('.')[0] in ('nn_layers', 'model', 'loss_fns', 'data_loader', 'utilities')]:
        del sys.modules[n]
    mspl_amd.install_dropin()
    from nn_layers.eesp import EESP, DownSampler  # noqa: F401
    from nn_layers.efficient_pyramid_pool import EfficientPyrPool  # noqa: F401
    from model.segmentation.espdnet_ue import espdnetue_seg2  # noqa: F401
    from data_loader.segmentation.greenhouse import id_camvid_to_greenhouse  # noqa: F401
    from loss_fns.segmentation_loss import (NIDLoss, PixelwiseKLD, SegmentationLoss,  # noqa: F401  (uest_seg_multi_os.py:36)
                                            UncertaintyWeightedSegmentationLoss)
    from utilities.metrics.segmentation_miou import MIOU  # noqa: F401  (uest_seg_multi_os.py:33)
    from nn_layers.aspp import ASPP, ASPP_Bottleneck  # noqa: F401  (model/segmentation/deeplabv3.py)
    from model.segmentation.espdnet import ESPDNetSegmentation, espdnet_seg, espdnet_seg_with_pre_rgbd  # noqa: F401
    from nn_layers.fusion_gate import FusionGate  # noqa: F401
    from mspl_amd import losses
    assert EESP is layers.EESP and espdnetue_seg2 is models.espdnetue_seg2 and PixelwiseKLD is losses.PixelwiseKLD
    # the callers' keyword (uest_seg_multi_os.py:509) and the in-place zeroing of the ignore class (:152-153)
    w = torch.ones(5)
    crit = UncertaintyWeightedSegmentationLoss(5, class_wts=w, ignore_idx=4, device='cpu')
    assert crit.class_weights[4] == 0.0 and crit.class_weights[:4].eq(1).all()
    assert NIDLoss(image_bin=16, label_bin=5).K == 16          # constructible without a GPU (uest_seg_multi_os.py:514)


def _purge_reference_names():
    import sys
    for n in [k for k in sys.modules if k.split('.')[0] in ('nn_layers', 'model', 'loss_fns', 'data_loader', 'utilities',
                                                             'transforms', '_mspl_reference')]:
        del sys.modules[n]


def test_dropin_overlays_real_modules(tmp_path, monkeypatch):
    """install_dropin() must not hide the other names of a module it aliases (uest_seg_multi_os.py:377 imports
    GreenhouseRGBDSegmentation and GREENHOUSE_CLASS_LIST from the module whose LUTs are aliased; :31-46, :556-559).  A
    throw-away package tree stands in for the reference: its own names pass through, conflicting names lose to the HIP-backed
    ones, and a real module is not executed until one of its own names is asked for."""
    import sys
    tree = {
        'data_loader/__init__.py': '',
        'data_loader/segmentation/__init__.py': '',
        'data_loader/segmentation/greenhouse.py':
            'import builtins\nbuiltins._mspl_fake_greenhouse_runs = getattr(builtins, "_mspl_fake_greenhouse_runs", 0) + 1\n'
            'GREENHOUSE_CLASS_LIST = ["a", "b", "c"]\nid_camvid_to_greenhouse = "CONFLICT"\n'
            'class GreenhouseRGBDSegmentation:\n    tag = "fake dataset"\n',
        'data_loader/segmentation/camvid.py': 'CAMVID = 1\n',
        'nn_layers/__init__.py': '',
        'nn_layers/espnet_utils.py': 'class CDilatedB:\n    pass\nclass CBR:\n    fake = True\n',
        # a real module that star-imports an aliased one (nn_layers/eesp.py:3 does): sees HIP names and pass-through names
        'nn_layers/eesp.py': 'from nn_layers.espnet_utils import *\nclass EESP:\n    fake = True\n'
                             'SEEN = (CBR, CDilatedB)\nOTHER = 7\n',
        'nn_layers/untouched.py': 'from nn_layers.espnet_utils import CBR, CDilatedB\n',
        'loss_fns/__init__.py': '',
        'loss_fns/segmentation_loss.py': 'class SelectiveBCE:\n    pass\nclass SoftArgMax:\n    pass\n',
    }
    for rel, src in tree.items():
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(src)
    import builtins
    monkeypatch.setattr(builtins, '_mspl_fake_greenhouse_runs', 0, raising=False)
    monkeypatch.syspath_prepend(str(tmp_path))
    _purge_reference_names()
    try:
        mspl_amd.install_dropin()
        from data_loader.segmentation.greenhouse import id_camvid_to_greenhouse, id_forest_to_greenhouse
        assert id_camvid_to_greenhouse is uest.id_camvid_to_greenhouse and id_forest_to_greenhouse is uest.id_forest_to_greenhouse
        assert builtins._mspl_fake_greenhouse_runs == 0, 'the real module ran although only aliased names were imported'
        from data_loader.segmentation.greenhouse import GREENHOUSE_CLASS_LIST, GreenhouseRGBDSegmentation   # :377
        assert GREENHOUSE_CLASS_LIST == ['a', 'b', 'c'] and GreenhouseRGBDSegmentation.tag == 'fake dataset'
        assert builtins._mspl_fake_greenhouse_runs == 1
        import data_loader.segmentation.greenhouse as gh
        assert gh.id_camvid_to_greenhouse is uest.id_camvid_to_greenhouse           # the HIP-side name wins the conflict
        from data_loader.segmentation.greenhouse import GREENHOUSE_CLASS_LIST as again   # noqa: F401
        assert builtins._mspl_fake_greenhouse_runs == 1, 'the real module must load once'
        from data_loader.segmentation.camvid import CAMVID              # sibling modules stay importable
        assert CAMVID == 1
        from nn_layers.espnet_utils import CBR, CDilatedB
        assert CBR is layers.CBR and CDilatedB.__module__ == '_mspl_reference.nn_layers.espnet_utils'
        import nn_layers.eesp as eesp
        assert eesp.EESP is layers.EESP and eesp.OTHER == 7
        assert eesp.SEEN[0] is layers.CBR and eesp.SEEN[1] is CDilatedB   # the real module's star import goes through the alias
        import nn_layers.untouched                                         # noqa: F401  (a non-aliased real module, same imports)
        from loss_fns.segmentation_loss import PixelwiseKLD, SelectiveBCE, SoftArgMax   # noqa: F401
        from mspl_amd import losses
        assert PixelwiseKLD is losses.PixelwiseKLD
        with pytest.raises(ImportError):
            from loss_fns.segmentation_loss import NoSuchLoss              # noqa: F401
        with pytest.raises(AttributeError):
            gh.no_such_name
        from model.segmentation.espdnet_ue import espdnetue_seg            # espdnet_ue.py:384; no real `model` package here
        assert espdnetue_seg is models.espdnetue_seg
        with pytest.raises(ImportError):
            from model.segmentation.espdnet_ue import something_else       # noqa: F401
    finally:
        _purge_reference_names()
        if hasattr(builtins, '_mspl_fake_greenhouse_runs'):
            monkeypatch.delattr(builtins, '_mspl_fake_greenhouse_runs')


def test_patch_script_binds_the_script_level_functions():
    """The five functions uest_seg_multi_os.py defines itself (:669-956) are rebound in the script's namespace; the adapters keep
    the reference's positional signatures (:730-731, :832-833)."""
    import inspect
    from mspl_amd import io as mio, script
    ns = {'main': 1}
    assert script.patch_script(ns) == ['generate_pseudo_label', 'generate_pseudo_label_multi_model', 'get_output', 'merge_outputs',
                                       'update_image_list']
    assert ns['main'] == 1 and ns['get_output'] is uest.get_output and ns['update_image_list'] is mio.update_image_list
    ref_tail = ['device', 'save_path', 'round_idx', 'tgt_num', 'label_2_id', 'valid_labels', 'args', 'logger', 'class_encoding', 'writer']
    assert list(inspect.signature(ns['generate_pseudo_label']).parameters)[:11] == ['model'] + ref_tail
    assert list(inspect.signature(ns['generate_pseudo_label_multi_model']).parameters)[:12] == ['model_list', 'os_data_list'] + ref_tail
    import argparse
    with pytest.raises(RuntimeError, match='greenhouse'):
        ns['generate_pseudo_label'](None, 'cuda', '/nonexistent', 0, args=argparse.Namespace(eval_training=False, classes=5, dataset='camvid'))


def test_espdnetue_seg_loader(tmp_path):
    """espdnet_ue.py:384-452: the older factory (pyr_plane_proj = min(classes//2, 16); whole-file load without a shape filter)."""
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000, classes=5, dataset='greenhouse', dense_fuse=False,
                           trainable_fusion=True, weights='')
    m = models.espdnetue_seg(a)
    assert m.bu_dec_l1.proj_planes == 2                   # min(5 // 2, 16)
    sd = synth_state_dict(m.state_dict(), 5)
    path = str(tmp_path / 'ck.pth')
    torch.save(sd, path)
    a.weights = path
    m2 = models.espdnetue_seg(a, load_entire_weights=True)
    got = m2.state_dict()
    assert torch.equal(got['bu_dec_l1.merge_layer.3.weight'], sd['bu_dec_l1.merge_layer.3.weight'])
    assert torch.equal(got['depth_base_net.level1.conv.weight'], sd['base_net.level1.conv.weight'].mean(1, keepdim=True))
    torch.save({k: v for k, v in sd.items() if k != 'base_net.level1.conv.weight'}, path)
    with pytest.raises(KeyError):
        models.espdnetue_seg(a)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'mspl_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f
                assert 'oracle.' not in src.replace('the oracle', ''), f


@pytest.mark.skipif(torch.cuda.is_available(), reason='CPU-only behaviour')
def test_cpu_tensors_fail_loudly():
    m = layers.EESP(32, 32, stride=1, k=4, r_lim=9).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match='no CPU path'):
        m(torch.randn(1, 32, 8, 8))


def test_training_mode_bn_fails_loudly():
    m = layers.CBR(3, 8, 3, 2)   # fresh modules are in training mode
    with torch.no_grad(), pytest.raises(RuntimeError, match='training mode'):
        m(torch.randn(1, 3, 8, 8))


def test_aspp_state_dict_matches_reference_tables():
    """ASPP / ASPP_Bottleneck containers: same keys and shapes as nn_layers/aspp.py (tables written by make_golden.py)."""
    from mspl_amd import aspp
    from tests.cases import ASPP_CASES
    tables = json.load(open(os.path.join(GOLDEN, 'aspp_keys.json')))
    for name, (cls, ncls, _, _, _) in ASPP_CASES.items():
        m = getattr(aspp, cls)(num_classes=ncls)
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == tables[name]


def test_lr_schedule_helpers():
    """lr_poly / adjust_learning_rate (uest_seg_multi_os.py:1317-1328) against the formula, through the param_groups spelling."""
    from mspl_amd import training

    class Opt:
        param_groups = [{'lr': 0.0}]
    o = Opt()
    assert training.lr_poly(5e-4, 10, 100, 0.0) == 5e-4
    assert abs(training.lr_poly(1e-3, 25, 100, 0.9) - 1e-3 * 0.75 ** 0.9) < 1e-15
    assert training.adjust_learning_rate(o, 50, 100, 2e-3, 2.0) == o.param_groups[0]['lr'] == 2e-3 * 0.25


def test_espdnet_state_dict_and_loaders(tmp_path):
    """ESPDNetSegmentation: the reference's keys/shapes (table written by make_golden.py) and the two factory loaders'
    key-selection rules (model/segmentation/espdnet.py:312-417)."""
    from tests.cases import ESPDNET_CASES
    tables = json.load(open(os.path.join(GOLDEN, 'espdnet_keys.json')))
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000, classes=5, dataset='greenhouse', dense_fuse=False,
                           trainable_fusion=True, weights='')
    for name, (classes, dataset, _, _, _, _, dense, trainable) in ESPDNET_CASES.items():
        m = models.ESPDNetSegmentation(a, classes=classes, dataset=dataset, dense_fuse=dense, trainable_fusion=trainable)
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == tables[name]
    with pytest.raises(KeyError):
        models.ESPDNetSegmentation(a, classes=5, dataset='forest')          # espdnet.py:91-99 has no 'forest' row
    # espdnet_seg: a classification-style file (keys without the 'base_net.' prefix) fills base_net AND depth_base_net
    src = models.EESPNet(a)
    sd = synth_state_dict(src.state_dict(), 77)
    f = tmp_path / 'cls.pth'
    torch.save(sd, str(f))
    a.weights = str(f)
    m = models.espdnet_seg(a)
    assert torch.equal(m.base_net.level3[1].proj_1x1.conv.weight, sd['level3.1.proj_1x1.conv.weight'])
    assert torch.equal(m.depth_base_net.level1.conv.weight, sd['level1.conv.weight'].mean(1, keepdim=True))
    assert torch.equal(m.depth_base_net.level4[6].conv_1x1_exp.conv.weight, sd['level4.6.conv_1x1_exp.conv.weight'])
    # espdnet_seg_with_pre_rgbd: whole-model file; decoder keys only with load_entire_weights
    full = synth_state_dict(m.state_dict(), 78)
    g = tmp_path / 'full.pth'
    torch.save(full, str(g))
    a.weights = str(g)
    m1 = models.espdnet_seg_with_pre_rgbd(a)
    m2 = models.espdnet_seg_with_pre_rgbd(a, load_entire_weights=True, ignore_layers=['bu_br_l2.1.weight'])
    k = 'bu_dec_l1.projection_layer.cbr.0.weight'
    assert torch.equal(m1.state_dict()['depth_base_net.level1.conv.weight'], full['depth_base_net.level1.conv.weight'])
    assert not torch.equal(m1.state_dict()[k], full[k]) and torch.equal(m2.state_dict()[k], full[k])
    assert not torch.equal(m2.state_dict()['bu_br_l2.1.weight'], full['bu_br_l2.1.weight'])


def test_settle_seating_decision_on_injected_timings():
    """GraphedTrainStep._settle_streams' decision (mspl_amd/training.py: settle_seating) on injected timings: lanes that overlap at
    once / only after a re-seating / never.  The replacement of DataParallel's implicit overlap (utilities/parallel_wrapper.py:17-45)
    must say so when the lanes do not run side by side instead of silently taking 1.6x the time."""
    from mspl_amd.training import SETTLE_THRESHOLD, settle_seating
    serial = 15.0

    def run(first, later, attempts=3):
        made = []

        def cand():
            made.append(len(made) + 1)
            return made[-1]

        d = settle_seating(serial, first, lambda k: later[k - 1], cand, attempts)
        return d, len(made)

    # overlapped at once: no candidate is ever built
    d, made = run(0.44 * serial, [])
    assert d == {'seating': 0, 'lanes_ms': 0.44 * serial, 'settled': True, 'attempts': 0} and made == 0
    # two lanes share a queue (0.7 of serial), the second candidate overlaps: stop there, do not build a third
    d, made = run(0.70 * serial, [0.72 * serial, 0.45 * serial, 0.40 * serial])
    assert d['seating'] == 2 and d['settled'] and d['attempts'] == 2 and made == 2 and d['lanes_ms'] == 0.45 * serial
    # never settles: bounded number of candidates, the fastest seating seen is kept and the outcome says "not settled"
    d, made = run(0.95 * serial, [0.9 * serial, 0.7 * serial, 0.8 * serial])
    assert made == 3 and d['attempts'] == 3 and not d['settled'] and d['seating'] == 2 and d['lanes_ms'] == 0.7 * serial
    # a slower candidate never replaces the first seating
    d, made = run(0.6 * serial, [0.9 * serial] * 3)
    assert d['seating'] == 0 and not d['settled']
    # exactly at the threshold counts as settled
    d, _ = run(SETTLE_THRESHOLD * serial, [])
    assert d['settled']


def test_product_library_reads_no_environment():
    """include/mspl_hip.h promises a library without global mutable state: the default build imports neither getenv (the ~70 tuning
    switches of rounds 1-4 are compiled to their defaults; `make TUNING=1` builds libmspl_hip_tuning.so for the probes) nor
    hipMalloc (it never allocates device memory; the stamp buffers exist in `make STAMPS=1` builds only)."""
    import subprocess
    undefined = subprocess.run(['nm', '-D', '--undefined-only', _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = {ln.split()[-1].split('@')[0] for ln in undefined.splitlines() if ln.strip()}
    assert 'getenv' not in names and 'secure_getenv' not in names
    assert not any(n.startswith('hipMalloc') for n in names)
    blob = open(_native.LIB_PATH, 'rb').read()
    assert b'MSPL_PW_' not in blob and b'MSPL_DW_' not in blob and b'MSPL_PYR_' not in blob
