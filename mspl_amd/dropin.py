"""Make the reference's import lines resolve to this package.

uest_seg_multi_os.py:31-46,402-409, utilities/utils.py:278-298 and train_segmentation.py:187-202 import
`nn_layers.*`, `model.segmentation.*`, `model.classification.*`, `loss_fns.segmentation_loss` and the LUTs
from `data_loader.segmentation.greenhouse`.  install_dropin() registers alias modules under those names so
the existing scripts pick up the HIP-backed classes without edits (call it before their imports run).
"""
import os
import sys
import types


def _alias(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    mod.__dict__['__mspl_dropin__'] = True
    if not attrs:
        # a parent package created on the way: keep the reference's other sub-modules importable through it
        # (utilities.utils, data_loader.segmentation.camvid, ...) by pointing __path__ at the real directories
        rel = os.path.join(*name.split('.'))
        mod.__path__ = [os.path.join(p or '.', rel) for p in sys.path if os.path.isdir(os.path.join(p or '.', rel))]
    sys.modules[name] = mod
    parent, _, leaf = name.rpartition('.')
    if parent:
        if parent not in sys.modules:
            _alias(parent)
        setattr(sys.modules[parent], leaf, mod)
    return mod


def install_dropin(force=False):
    """Register the alias modules.  Refuses to shadow already-imported reference modules unless force=True."""
    from . import layers as L, models as M, uest as U
    names = ['nn_layers', 'model', 'loss_fns']
    if not force:
        for n in names:
            m = sys.modules.get(n)
            if m is not None and not getattr(m, '__mspl_dropin__', False):
                raise RuntimeError('mspl_amd.install_dropin: module %r is already imported from %r' %
                                   (n, getattr(m, '__file__', '?')))
    _alias('nn_layers.espnet_utils', CBR=L.CBR, BR=L.BR, CB=L.CB, C=L.C, CDilated=L.CDilated)
    _alias('nn_layers.cnn_utils', CBR=L.DecCBR, BR=L.DecBR, Shuffle=L.Shuffle)
    _alias('nn_layers.eesp', EESP=L.EESP, DownSampler=L.DownSampler)
    _alias('nn_layers.efficient_pyramid_pool', EfficientPyrPool=L.EfficientPyrPool)
    _alias('nn_layers.efficient_pt', EfficientPWConv=L.EfficientPWConv)
    _alias('nn_layers.fusion_gate', FusionGate=M.FusionGate)
    _alias('model.classification.espnetv2', EESPNet=M.EESPNet)
    _alias('model.classification.espnetv2_config', sc_ch_dict=M.sc_ch_dict, rep_layers=M.rep_layers,
           recept_limit=M.recept_limit, branches=M.branches, config_inp_reinf=L.config_inp_reinf,
           input_reinforcement=M.input_reinforcement)
    _alias('model.segmentation.espdnet_ue', ESPDNetwithUncertaintyEstimation=M.ESPDNetwithUncertaintyEstimation,
           espdnetue_seg2=M.espdnetue_seg2)
    _alias('model.segmentation.espdnet', ESPDNetSegmentation=M.ESPDNetSegmentation, espdnet_seg=M.espdnet_seg,
           espdnet_seg_with_pre_rgbd=M.espdnet_seg_with_pre_rgbd)
    _alias('model.segmentation.espnetv2', ESPNetv2Segmentation=M.ESPNetv2Segmentation, espnetv2_seg=M.espnetv2_seg)
    _alias('data_loader.segmentation.greenhouse', id_camvid_to_greenhouse=U.id_camvid_to_greenhouse,
           id_cityscapes_to_greenhouse=U.id_cityscapes_to_greenhouse, id_forest_to_greenhouse=U.id_forest_to_greenhouse)
    from . import aspp as A
    _alias('nn_layers.aspp', ASPP=A.ASPP, ASPP_Bottleneck=A.ASPP_Bottleneck)
    from . import losses as S
    from . import lr_scheduler as R
    _alias('utilities.lr_scheduler', CyclicLR=R.CyclicLR, FixedMultiStepLR=R.FixedMultiStepLR, PolyLR=R.PolyLR,
           LinearLR=R.LinearLR, HybirdLR=R.HybirdLR, CosineLR=R.CosineLR)
    from . import metrics as Q
    _alias('utilities.metrics.segmentation_miou', MIOU=Q.MIOU)
    _alias('loss_fns.segmentation_loss', PixelwiseKLD=S.PixelwiseKLD,
           UncertaintyWeightedSegmentationLoss=S.UncertaintyWeightedSegmentationLoss,
           SegmentationLoss=S.SegmentationLoss, NIDLoss=S.NIDLoss)
