"""Make the reference's import lines resolve to this package.

uest_seg_multi_os.py:31-46,402-409, utilities/utils.py:278-298 and train_segmentation.py:187-202 import
`nn_layers.*`, `model.segmentation.*`, `model.classification.*`, `loss_fns.segmentation_loss` and the LUTs
from `data_loader.segmentation.greenhouse`.  install_dropin() registers alias modules under those names so
the existing scripts pick up the HIP-backed classes without edits (call it before their imports run).

The aliases are OVERLAYS, not replacements: a name the alias does not define (`GreenhouseRGBDSegmentation`,
`GREENHOUSE_CLASS_LIST`, `CDilatedB`, `cnn_utils.CB`, `SelectiveBCE`, ... -- uest_seg_multi_os.py:377,556-559) is
forwarded to the real module of the same dotted name when one is importable from sys.path.  That module is loaded
under a private name (`_mspl_reference.<name>`) the first time such a name is asked for and never before, so
scripts that only use the HIP-backed names never execute it (the real `data_loader.segmentation.greenhouse` imports
cv2 / torchvision at the top).  Names the alias defines always win.  Without a real module on sys.path the alias
behaves like a plain module (AttributeError / ImportError for unknown names).
"""
import importlib.machinery
import importlib.util
import os
import sys
import types

_PRIVATE_PREFIX = '_mspl_reference.'
_MISSING = object()


def _real_dirs(name):
    rel = os.path.join(*name.split('.'))
    return [os.path.join(p or '.', rel) for p in sys.path if os.path.isdir(os.path.join(p or '.', rel))]


class _Overlay(types.ModuleType):
    """Alias module: own names first, then the same-named real module from sys.path (loaded lazily, privately)."""

    def _mspl_real(self):
        real = self.__dict__.get('__mspl_real__', _MISSING)
        if real is not _MISSING:
            return real
        real = None
        parent, _, leaf = self.__name__.rpartition('.')
        search = _real_dirs(parent) if parent else None
        spec = importlib.machinery.PathFinder.find_spec(leaf, search)
        if spec is not None and spec.loader is not None and spec.origin and os.path.isfile(spec.origin):
            private = _PRIVATE_PREFIX + self.__name__
            spec = importlib.util.spec_from_file_location(private, spec.origin)
            real = importlib.util.module_from_spec(spec)
            # the real module's own `from nn_layers.x import ...` lines resolve to the aliases (HIP-backed names win there too)
            sys.modules[private] = real
            self.__dict__['__mspl_real__'] = real         # set before exec: a circular import sees the partial module
            try:
                spec.loader.exec_module(real)
            except BaseException:
                sys.modules.pop(private, None)
                self.__dict__.pop('__mspl_real__', None)
                raise
        self.__dict__['__mspl_real__'] = real
        return real

    def __getattr__(self, name):
        if name == '__all__':                             # `from alias import *`: own public names + the real module's
            own = [k for k in self.__dict__ if not k.startswith('_')]
            real = self._mspl_real()
            if real is not None:
                pub = getattr(real, '__all__', None) or [k for k in real.__dict__ if not k.startswith('_')]
                own += [k for k in pub if k not in self.__dict__]
            return own
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        real = self._mspl_real()
        if real is not None and hasattr(real, name):
            return getattr(real, name)
        raise AttributeError('module %r has no attribute %r (mspl_amd drop-in alias%s)' % (
            self.__name__, name, '' if real is not None else '; no reference module of that name on sys.path'))


def _alias(name, **attrs):
    mod = _Overlay(name) if attrs else types.ModuleType(name)
    mod.__dict__.update(attrs)
    mod.__dict__['__mspl_dropin__'] = True
    if not attrs:
        # a parent package created on the way: keep the reference's other sub-modules importable through it
        # (utilities.utils, data_loader.segmentation.camvid, ...) by pointing __path__ at the real directories
        mod.__path__ = _real_dirs(name)
    sys.modules[name] = mod
    parent, _, leaf = name.rpartition('.')
    if parent:
        if parent not in sys.modules:
            _alias(parent)
        setattr(sys.modules[parent], leaf, mod)
    return mod


def install_dropin(force=False, script=None):
    """Register the alias modules.  Refuses to shadow already-imported reference modules unless force=True.
    script: the globals() (or module object) of uest_seg_multi_os.py -- the functions the script defines itself (get_output,
    merge_outputs, update_image_list, generate_pseudo_label, generate_pseudo_label_multi_model) are rebound there too
    (mspl_amd.script.patch_script; call after the script's own definitions)."""
    from . import layers as L, models as M, uest as U
    names = ['nn_layers', 'model', 'loss_fns']
    if not force:
        for n in names:
            m = sys.modules.get(n)
            if m is not None and not getattr(m, '__mspl_dropin__', False):
                raise RuntimeError('mspl_amd.install_dropin: module %r is already imported from %r' %
                                   (n, getattr(m, '__file__', '?')))
    for n in [k for k in sys.modules if k.startswith(_PRIVATE_PREFIX)]:
        del sys.modules[n]
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
           espdnetue_seg2=M.espdnetue_seg2, espdnetue_seg=M.espdnetue_seg)
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
    from . import evaluation as E
    _alias('utilities.train_eval_seg', val_seg_ue=E.val_seg_ue)
    _alias('loss_fns.segmentation_loss', PixelwiseKLD=S.PixelwiseKLD,
           UncertaintyWeightedSegmentationLoss=S.UncertaintyWeightedSegmentationLoss,
           SegmentationLoss=S.SegmentationLoss, NIDLoss=S.NIDLoss)
    if script is not None:
        from .script import patch_script
        patch_script(script)
