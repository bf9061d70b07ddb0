"""Segmentation networks with the reference's names, constructor arguments, state-dict keys and quirks.

Reference surface mirrored (paths relative to the reference root):
  model/classification/espnetv2.py:15-142         EESPNet (encoder; classifier/level5 dropped by the seg nets)
  model/classification/espnetv2_config.py:6-23    channel tables, repetition counts, receptive-field limits
  model/segmentation/espdnet_ue.py:18-302         ESPDNetwithUncertaintyEstimation, :304-382 espdnetue_seg2
  model/segmentation/espnetv2.py:16-198           ESPNetv2Segmentation, espnetv2_seg
  model/segmentation/espdnet.py:18-417            ESPDNetSegmentation, espdnet_seg, espdnet_seg_with_pre_rgbd
  nn_layers/fusion_gate.py:11-47                  FusionGate (RGB-D fusion, x_d path of espdnet_ue.py:186-270)

Beyond the reference API each network offers `forward_lowres(x)` -- the decoder outputs before the final
bilinear upsample -- which the pseudo-label pass feeds to the fused label epilogue so that full-resolution
logits are never written to HBM.
"""
import copy
import os

import torch
from torch import nn
from torch.nn import init

from . import ops
from . import autograd as ag
from .layers import (C, CBR, DownSampler, EESP, EfficientPWConv, EfficientPyrPool, ImagePyramid, _training_path, cached, run_eesp_chain,
                     decoder_merge, fork, join, wait_mark)

sc_ch_dict = {
    0.5: [16, 32, 64, 128, 256, 1024],
    1.0: [32, 64, 128, 256, 512, 1024],
    1.25: [32, 80, 160, 320, 640, 1024],
    1.5: [32, 96, 192, 384, 768, 1024],
    2.0: [32, 128, 256, 512, 1024, 1280],
}
rep_layers = [0, 3, 7, 3]
recept_limit = [13, 11, 9, 7, 5]
branches = 4
input_reinforcement = True

DEC_FEAT = {'pascal': 16, 'city': 16, 'coco': 32, 'greenhouse': 16, 'ishihara': 16, 'sun': 16, 'camvid': 16,
            'forest': 16}


def _init_params(model):
    """init_params of the reference models (espdnet_ue.py:112-127)."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            init.kaiming_normal_(m.weight, mode='fan_out')
            if m.bias is not None:
                init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm2d):
            init.constant_(m.weight, 1)
            init.constant_(m.bias, 0)
        elif isinstance(m, nn.Linear):
            init.normal_(m.weight, std=0.001)
            if m.bias is not None:
                init.constant_(m.bias, 0)


def _param_gen(modules):
    for mod in modules:
        for _, m in mod.named_modules():
            if isinstance(m, (nn.Conv2d, nn.BatchNorm2d, nn.PReLU)):
                for p in m.parameters():
                    if p.requires_grad:
                        yield p


class EESPNet(nn.Module):
    """Encoder trunk.  Only the levels the segmentation nets keep (1-4) are built; the ImageNet head
    (level5, classifier) is deleted by the reference right after construction (espdnet_ue.py:35-37)."""

    def __init__(self, args):
        super().__init__()
        channels_in = getattr(args, 'channels', 3)
        s = args.s
        if s not in sc_ch_dict:
            raise ValueError('Model at scale s={} is not supported yet'.format(s))
        cm = sc_ch_dict[s]
        K = [branches] * len(recept_limit)
        self.input_reinforcement = getattr(args, 'input_reinforcement', input_reinforcement)
        self.level1 = CBR(channels_in, cm[0], 3, 2)
        self.level2_0 = DownSampler(cm[0], cm[1], k=K[0], r_lim=recept_limit[0], reinf=self.input_reinforcement)
        self.level3_0 = DownSampler(cm[1], cm[2], k=K[1], r_lim=recept_limit[1], reinf=self.input_reinforcement)
        self.level3 = nn.ModuleList(EESP(cm[2], cm[2], stride=1, k=K[2], r_lim=recept_limit[2])
                                    for _ in range(rep_layers[1]))
        self.level4_0 = DownSampler(cm[2], cm[3], k=K[2], r_lim=recept_limit[2], reinf=self.input_reinforcement)
        self.level4 = nn.ModuleList(EESP(cm[3], cm[3], stride=1, k=K[3], r_lim=recept_limit[3])
                                    for _ in range(rep_layers[2]))
        self.config = cm
        _init_params(self)


class FusionGate(nn.Module):
    """RGB-D fusion gate (nn_layers/fusion_gate.py:11-47): out = rgb*w + depth*(1-w), w = sigmoid(conv_1x1(cat(rgb, depth))),
    or rgb + depth when not trainable.  The concatenation is never materialised on the inference path: the 1x1 runs as two
    matrix-core launches over the two halves of the weight (the second accumulates onto the first), then one blend kernel.
    The reference's hard-coded `.to('cuda')` (:38) has no counterpart: everything stays on the inputs' device."""

    def __init__(self, nchannel, is_trainable=True):
        super().__init__()
        self.nchannel = nchannel
        self.conv_1x1 = C(nIn=2 * nchannel, nOut=nchannel, kSize=1)
        self.sigmoid = nn.Sigmoid()
        self.is_trainable = is_trainable

    def forward(self, rgb, depth):
        if rgb.shape != depth.shape or rgb.shape[1] != self.nchannel:
            raise RuntimeError('mspl_amd: FusionGate(%d) got rgb %s and depth %s'
                               % (self.nchannel, tuple(rgb.shape), tuple(depth.shape)))
        w = self.conv_1x1.conv.weight
        if _training_path():
            z = ag.conv(torch.cat((rgb, depth), 1), w) if self.is_trainable else None
            return ag.FusionGateFn.apply(z, rgb, depth)
        if not self.is_trainable:
            return ops.fusion_gate(None, rgb, depth)
        c = self.nchannel
        w_rgb, w_d = cached(self, 'halves', [w], lambda: (w[:, :c].contiguous(), w[:, c:].contiguous()))
        z = ops.conv1x1(rgb, w_rgb)
        z = ops.conv1x1(depth, w_d, ep=ops.Epi(pre_add=z))
        return ops.fusion_gate(z, rgb, depth)


class _SegBase(nn.Module):
    def _build_decoder(self, config, classes, dataset, pyr_plane_proj, with_aux, aux_layer=2):
        base = DEC_FEAT[dataset]
        dec = [4 * base, 3 * base, 2 * base, classes]
        self.bu_dec_l1 = EfficientPyrPool(in_planes=config[3], proj_planes=pyr_plane_proj, out_planes=dec[0])
        self.bu_dec_l2 = EfficientPyrPool(in_planes=dec[0], proj_planes=pyr_plane_proj, out_planes=dec[1])
        self.bu_dec_l3 = EfficientPyrPool(in_planes=dec[1], proj_planes=pyr_plane_proj, out_planes=dec[2])
        self.bu_dec_l4 = EfficientPyrPool(in_planes=dec[2], proj_planes=pyr_plane_proj, out_planes=dec[3],
                                          last_layer_br=False)
        self.merge_enc_dec_l2 = EfficientPWConv(config[2], dec[0])
        self.merge_enc_dec_l3 = EfficientPWConv(config[1], dec[1])
        self.merge_enc_dec_l4 = EfficientPWConv(config[0], dec[2])
        self.bu_br_l2 = nn.Sequential(nn.BatchNorm2d(dec[0]), nn.PReLU(dec[0]))
        self.bu_br_l3 = nn.Sequential(nn.BatchNorm2d(dec[1]), nn.PReLU(dec[1]))
        self.bu_br_l4 = nn.Sequential(nn.BatchNorm2d(dec[2]), nn.PReLU(dec[2]))
        if with_aux:
            self.aux_layer = aux_layer
            if 0 <= aux_layer < 3:
                self.aux_decoder = EfficientPyrPool(in_planes=dec[aux_layer], proj_planes=pyr_plane_proj,
                                                    out_planes=dec[3], last_layer_br=False)

    def _encode(self, x, image_for_l2, l3_tail, pyr=None):
        b = self.base_net
        if not b.input_reinforcement:
            pyr = None
        elif pyr is None:
            pyr = ImagePyramid(x.detach())
        l1 = b.level1(x)
        if _training_path():
            # every encoder output has four consumers (the next DownSampler's pool and EESP, the skip connection's gate and 3x3):
            # four aliases whose gradients ONE launch sums (autograd.fan_out) instead of three pairwise ATen adds per level
            a1 = ag.fan_out(l1, 4)
            l2 = b.level2_0(a1[0], pyr if image_for_l2 else None, _alias=a1[1])
            a2 = ag.fan_out(l2, 4)
            l3 = b.level3_0(a2[0], pyr, _alias=a2[1])
            for i, layer in enumerate(b.level3):
                l3 = (layer if i == 0 else l3_tail[i])(l3)
            a3 = ag.fan_out(l3, 4)
            l4 = b.level4_0(a3[0], pyr, _alias=a3[1])
            for layer in b.level4:
                l4 = layer(l4)
            return a1[2:], a2[2:], a3[2:], l4
        l2 = b.level2_0(l1, pyr if image_for_l2 else None)
        l3 = b.level3_0(l2, pyr)
        l3 = run_eesp_chain([layer if i == 0 else l3_tail[i] for i, layer in enumerate(b.level3)], l3)
        l4 = b.level4_0(l3, pyr)
        l4 = run_eesp_chain(b.level4, l4)
        return l1, l2, l3, l4

    def _encode_rgbd(self, x, x_d):
        """Encoder with the depth branch (espdnet_ue.py:186-270 == espdnet.py:200-284): depth features from depth_base_net
        (DownSamplers without image reinforcement), a FusionGate after level1, level2 and after the level3 / level4
        stacks (after every block with dense_fuse); the RGB branch's level3[1:] run through depth_base_net's blocks."""
        if x_d.dim() != 4 or x_d.shape[0] != x.shape[0] or x_d.shape[1] != 1 or x_d.shape[2:] != x.shape[2:]:
            raise RuntimeError('mspl_amd: depth input %s does not match image %s (expected (N,1,H,W))'
                               % (tuple(x_d.shape), tuple(x.shape)))
        b, d = self.base_net, self.depth_base_net
        pyr = ImagePyramid(x.detach()) if b.input_reinforcement else None
        dl1 = d.level1(x_d)                                                     # :187
        l1 = self.fusion_gate_level1(b.level1(x), dl1)                          # :191
        dl2 = d.level2_0(dl1)                                                   # :198 (no image reinforcement)
        l2 = self.fusion_gate_level2(b.level2_0(l1, pyr), dl2)                  # :196,:202
        l3 = b.level3_0(l2, pyr)                                                # :207
        dl3 = d.level3_0(dl2)                                                   # :209
        for i, (layer, dlayer) in enumerate(zip(b.level3, d.level3)):           # :213-228
            l3 = (layer if i == 0 else dlayer)(l3)
            dl3 = dlayer(dl3)
            if self.dense_fuse:
                l3 = self.fusion_gate_level3(l3, dl3)
        if not self.dense_fuse:
            l3 = self.fusion_gate_level3(l3, dl3)                               # :230-233
        l4 = b.level4_0(l3, pyr)                                                # :238
        dl4 = d.level4_0(dl3)                                                   # :240
        for layer, dlayer in zip(b.level4, d.level4):                           # :244-259
            l4 = layer(l4)
            dl4 = dlayer(dl4)
            if self.dense_fuse:
                l4 = self.fusion_gate_level4(l4, dl4)
        if not self.dense_fuse:
            l4 = self.fusion_gate_level4(l4, dl4)                               # :261-264
        return l1, l2, l3, l4

    def _decode(self, l1, l2, l3, l4, aux_layer):
        aux = None
        # the three skip connections depend on the encoder only, the auxiliary head on one decoder stage only: both
        # are issued on side streams (layers.fork / join) and overlap the main decoder chain
        if isinstance(l1, tuple):                                # training path: (gate alias, 3x3 alias) pairs from _encode's fan_out
            pw2 = self.merge_enc_dec_l2(l3[0], _alias=l3[1])
            pw3 = self.merge_enc_dec_l3(l2[0], _alias=l2[1])
            pw4 = self.merge_enc_dec_l4(l1[0], _alias=l1[1])
            e2 = e3 = e4 = None
            l1, l2, l3 = l1[0], l2[0], l3[0]
        else:
          with fork(1, (l1, l2, l3)) as f:
            pw2 = self.merge_enc_dec_l2(l3);  e2 = f.mark()      # each skip connection is awaited on its own event:
            pw3 = self.merge_enc_dec_l3(l2);  e3 = f.mark()      # stage k of the decoder starts when pw_k is ready, not
            pw4 = self.merge_enc_dec_l4(l1);  e4 = f.mark()      # when the whole side stream has drained
        bu = self.bu_dec_l1(l4)
        if aux_layer == 0:
            with fork(2, (bu,)):
                aux = self.aux_decoder(bu)
        wait_mark(e2, (pw2,))
        bu = decoder_merge(pw2, bu, self.bu_br_l2)
        bu = self.bu_dec_l2(bu)
        if aux_layer == 1:
            with fork(2, (bu,)):
                aux = self.aux_decoder(bu)
        wait_mark(e3, (pw3,))
        bu = decoder_merge(pw3, bu, self.bu_br_l3)
        bu = self.bu_dec_l3(bu)
        if aux_layer == 2:
            with fork(2, (bu,)):
                aux = self.aux_decoder(bu)
        wait_mark(e4, (pw4,))                                     # the side stream's last work: it is joined here
        bu = decoder_merge(pw4, bu, self.bu_br_l4)
        bu = self.bu_dec_l4(bu)
        join(2, (aux,))
        return bu, aux

    def get_basenet_params(self):
        return _param_gen([self.base_net])

    def get_segment_params(self):
        return _param_gen([self.bu_dec_l1, self.bu_dec_l2, self.bu_dec_l3, self.bu_dec_l4, self.merge_enc_dec_l4,
                           self.merge_enc_dec_l3, self.merge_enc_dec_l2, self.bu_br_l4, self.bu_br_l3, self.bu_br_l2])

    def upsample(self, x):
        return ops.bilinear(x, (x.shape[2] * 2, x.shape[3] * 2))


def _check_input(x):
    if x.dim() != 4:
        raise RuntimeError('mspl_amd: expected an (N,C,H,W) batch, got %s' % (tuple(x.shape),))
    if x.shape[2] % 16 or x.shape[3] % 16:
        # the reference fails at the first decoder skip-add; fail up front with its message shape
        raise RuntimeError('The size of tensor a must match the size of tensor b: input %dx%d is not a multiple of 16 '
                           '(mspl_amd: H/W not multiple of 16)' % (x.shape[2], x.shape[3]))


class ESPDNetwithUncertaintyEstimation(_SegBase):
    def __init__(self, args, classes=21, dataset='pascal', dense_fuse=False, trainable_fusion=True, aux_layer=2,
                 fix_pyr_plane_proj=False):
        super().__init__()
        self.base_net = EESPNet(args)
        config = self.base_net.config
        tmp_args = copy.deepcopy(args)
        tmp_args.channels = 1
        self.depth_base_net = EESPNet(tmp_args)
        self.fusion_gate_level1 = FusionGate(nchannel=32, is_trainable=trainable_fusion)
        self.fusion_gate_level2 = FusionGate(nchannel=128, is_trainable=trainable_fusion)
        self.fusion_gate_level3 = FusionGate(nchannel=256, is_trainable=trainable_fusion)
        self.fusion_gate_level4 = FusionGate(nchannel=512, is_trainable=trainable_fusion)
        base = DEC_FEAT[dataset]
        pyr = base if fix_pyr_plane_proj else min(classes // 2, base)
        self._build_decoder(config, classes, dataset, pyr, True, aux_layer)
        _init_params(self)
        self.dense_fuse = dense_fuse
        self.classes = classes

    def get_depth_encoder_params(self):
        return _param_gen([self.depth_base_net])

    def get_classification_layer_params(self):
        return self.get_segment_params()

    def forward_lowres(self, x, x_d=None, pyr=None):
        """(main at H/2 x W/2, aux at H/4 x W/4 for aux_layer=2): decoder outputs before espdnet_ue.py:301-302.
        pyr: an ImagePyramid of x shared by several models looking at the same batch (the multi-source label pass).
        With x_d the encoder follows espdnet_ue.py:186-270 line by line: depth features come from depth_base_net (its
        DownSamplers get NO image reinforcement, :198,:209,:240), fusion after level1, level2 and after the level3 /
        level4 stacks (after every block with dense_fuse)."""
        _check_input(x)
        if x_d is None:
            # level3[1:] run through depth_base_net's layers (espdnet_ue.py:226) -- reproduced on purpose
            l1, l2, l3, l4 = self._encode(x, True, self.depth_base_net.level3, pyr)
            return self._decode(l1, l2, l3, l4, self.aux_layer)
        l1, l2, l3, l4 = self._encode_rgbd(x, x_d)
        return self._decode(l1, l2, l3, l4, self.aux_layer)

    def forward(self, x, x_d=None):
        main, aux = self.forward_lowres(x, x_d)
        if _training_path():
            return ag.bilinear(main, tuple(x.shape[2:])), ag.bilinear(aux, tuple(x.shape[2:]))
        r = ops.label_epilogue(main, aux, x.shape[2:], want_labels=False, want_logits=True)
        return r['main_up'], r['aux_up']


class ESPDNetSegmentation(_SegBase):
    """Single-head RGB(-D) network (model/segmentation/espdnet.py:18-309, `--os-model espdnet`): the ESPDNet-UE topology
    without the auxiliary decoder; pyr_plane_proj = min(classes//2, 16|32) (:101); forward returns one logits tensor."""

    def __init__(self, args, classes=21, dataset='pascal', dense_fuse=False, trainable_fusion=True):
        super().__init__()
        if dataset == 'forest':
            raise KeyError('forest')                      # espdnet.py:91-99: the table has no 'forest' entry
        self.base_net = EESPNet(args)
        config = self.base_net.config
        tmp_args = copy.deepcopy(args)
        tmp_args.channels = 1
        self.depth_base_net = EESPNet(tmp_args)
        self.fusion_gate_level1 = FusionGate(nchannel=32, is_trainable=trainable_fusion)
        self.fusion_gate_level2 = FusionGate(nchannel=128, is_trainable=trainable_fusion)
        self.fusion_gate_level3 = FusionGate(nchannel=256, is_trainable=trainable_fusion)
        self.fusion_gate_level4 = FusionGate(nchannel=512, is_trainable=trainable_fusion)
        self._build_decoder(config, classes, dataset, min(classes // 2, DEC_FEAT[dataset]), False)
        _init_params(self)
        self.dense_fuse = dense_fuse
        self.classes = classes

    def get_depth_encoder_params(self):
        return _param_gen([self.depth_base_net])

    def forward_lowres(self, x, x_d=None):
        _check_input(x)
        if x_d is None:
            l1, l2, l3, l4 = self._encode(x, True, self.depth_base_net.level3)      # espdnet.py:240: dlayer for i > 0
        else:
            l1, l2, l3, l4 = self._encode_rgbd(x, x_d)
        return self._decode(l1, l2, l3, l4, -1)[0], None

    def forward(self, x, x_d=None):
        main, _ = self.forward_lowres(x, x_d)
        if _training_path():
            return ag.bilinear(main, tuple(x.shape[2:]))
        return ops.bilinear(main, x.shape[2:])


class ESPNetv2Segmentation(_SegBase):
    def __init__(self, args, classes=21, dataset='pascal'):
        super().__init__()
        self.base_net = EESPNet(args)
        config = self.base_net.config
        self._build_decoder(config, classes, dataset, min(classes // 2, DEC_FEAT[dataset]), False)
        _init_params(self)
        self.classes = classes

    def forward_lowres(self, x):
        _check_input(x)
        # level2_0 is called WITHOUT the image (espnetv2.py:127)
        l1, l2, l3, l4 = self._encode(x, False, self.base_net.level3)
        return self._decode(l1, l2, l3, l4, -1)[0], None

    def forward(self, x):
        main, _ = self.forward_lowres(x)
        if _training_path():
            return ag.bilinear(main, tuple(x.shape[2:]))
        return ops.bilinear(main, x.shape[2:])


# ------------------------------------------------------------------ factories / checkpoint loaders
def _load_file(weights):
    if not os.path.isfile(weights):
        raise FileNotFoundError('Weight file does not exist at {}. Please check.'.format(weights))
    return torch.load(weights, map_location='cpu')


def espdnetue_seg2(args, load_entire_weights=False, fix_pyr_plane_proj=False):
    """espdnet_ue.py:304-382, including the lossy loader: after the main load, the SAME file refills
    depth_base_net from its base_net.* entries (k.lstrip('base_net.'), level1 averaged over RGB)."""
    model = ESPDNetwithUncertaintyEstimation(args, classes=args.classes, dataset=args.dataset,
                                             dense_fuse=args.dense_fuse, trainable_fusion=args.trainable_fusion,
                                             fix_pyr_plane_proj=fix_pyr_plane_proj)
    weights = args.weights
    if weights:
        pretrained = _load_file(weights)
        model_dict = model.state_dict()
        if load_entire_weights:
            overlap = {k: v for k, v in pretrained.items() if k in model_dict and model_dict[k].size() == v.size()}
            if len(overlap) == 0:
                raise RuntimeError('No overlaping weights between model file and pretrained weight file. Please check')
            model_dict.update(overlap)
            model.load_state_dict(model_dict)
        else:
            base_dict = model.base_net.state_dict()
            overlap = {k.replace('base_net.', ''): v for k, v in pretrained.items()
                       if k.replace('base_net.', '') in base_dict}
            if len(overlap) == 0:
                raise RuntimeError('No overlaping weights between model file and pretrained weight file. Please check')
            base_dict.update(overlap)
            model.base_net.load_state_dict(base_dict)
        # depth_weights = weights (espdnet_ue.py:310,361-380)
        d_dict = model.depth_base_net.state_dict()
        overlap = {k.lstrip('base_net.'): v for k, v in pretrained.items() if k.lstrip('base_net.') in d_dict}
        if 'level1.conv.weight' in overlap:
            overlap['level1.conv.weight'] = torch.mean(overlap['level1.conv.weight'], dim=1, keepdim=True)
        if len(overlap) == 0:
            raise RuntimeError('No overlaping weights between model file and pretrained weight file. Please check')
        d_dict.update(overlap)
        model.depth_base_net.load_state_dict(d_dict)
    return model


def espdnetue_seg(args, load_entire_weights=False):
    """espdnet_ue.py:384-452, the older loader kept beside espdnetue_seg2: pyr_plane_proj = min(classes//2, 16) (no
    fix_pyr_plane_proj), load_entire_weights takes every key the model has WITHOUT a shape check (a mismatching checkpoint
    raises in load_state_dict, like the reference), and the depth refill needs 'level1.conv.weight' (KeyError otherwise)."""
    model = ESPDNetwithUncertaintyEstimation(args, classes=args.classes, dataset=args.dataset,
                                             dense_fuse=args.dense_fuse, trainable_fusion=args.trainable_fusion)
    weights = args.weights
    if weights:
        pretrained = _load_file(weights)
        if load_entire_weights:
            model_dict = model.state_dict()
            overlap = {k: v for k, v in pretrained.items() if k in model_dict}
            if len(overlap) == 0:
                raise RuntimeError('No overlaping weights between model file and pretrained weight file. Please check')
            model_dict.update(overlap)
            model.load_state_dict(model_dict)
        else:
            base_dict = model.base_net.state_dict()
            overlap = {k.replace('base_net.', ''): v for k, v in pretrained.items()
                       if k.replace('base_net.', '') in base_dict}
            if len(overlap) == 0:
                raise RuntimeError('No overlaping weights between model file and pretrained weight file. Please check')
            base_dict.update(overlap)
            model.base_net.load_state_dict(base_dict)
        d_dict = model.depth_base_net.state_dict()
        overlap = {k.lstrip('base_net.'): v for k, v in pretrained.items() if k.lstrip('base_net.') in d_dict}
        overlap['level1.conv.weight'] = torch.mean(overlap['level1.conv.weight'], dim=1, keepdim=True)
        d_dict.update(overlap)
        model.depth_base_net.load_state_dict(d_dict)
    return model


def espnetv2_seg(args):
    """model/segmentation/espnetv2.py:170-198: only base_net.* keys are taken from the file."""
    model = ESPNetv2Segmentation(args, classes=args.classes, dataset=args.dataset)
    if args.weights:
        pretrained = _load_file(args.weights)
        base_dict = model.base_net.state_dict()
        overlap = {k: v for k, v in pretrained.items() if k in base_dict}
        if len(overlap) == 0:
            raise RuntimeError('No overlaping weights between model file and pretrained weight file. Please check')
        base_dict.update(overlap)
        model.base_net.load_state_dict(base_dict)
    return model


def espdnet_seg(args):
    """model/segmentation/espdnet.py:312-380: base_net takes the file's keys that match its own names; depth_base_net
    takes the same file's matching keys with level1.conv.weight averaged over RGB (KeyError, like the reference, when the
    file has no such key)."""
    model = ESPDNetSegmentation(args, classes=args.classes, dataset=args.dataset, dense_fuse=args.dense_fuse,
                                trainable_fusion=args.trainable_fusion)
    if args.weights:
        pretrained = _load_file(args.weights)
        base_dict = model.base_net.state_dict()
        overlap = {k: v for k, v in pretrained.items() if k in base_dict}
        if len(overlap) == 0:
            raise RuntimeError('No overlaping weights between model file and pretrained weight file. Please check')
        base_dict.update(overlap)
        model.base_net.load_state_dict(base_dict)
        d_dict = model.depth_base_net.state_dict()
        overlap = {k: v for k, v in pretrained.items() if k in d_dict}
        overlap['level1.conv.weight'] = torch.mean(overlap['level1.conv.weight'], dim=1, keepdim=True)
        d_dict.update(overlap)
        model.depth_base_net.load_state_dict(d_dict)
    return model


def espdnet_seg_with_pre_rgbd(args, ignore_layers=[], load_entire_weights=False):
    """model/segmentation/espdnet.py:382-417: keys present in the model, not ignored, and ('base_net' in the key or
    load_entire_weights)."""
    model = ESPDNetSegmentation(args, classes=args.classes, dataset=args.dataset, dense_fuse=args.dense_fuse,
                                trainable_fusion=args.trainable_fusion)
    if args.weights:
        pretrained = _load_file(args.weights)
        model_dict = model.state_dict()
        overlap = {k: v for k, v in pretrained.items()
                   if (k in model_dict) and k not in ignore_layers and ('base_net' in k or load_entire_weights)}
        if len(overlap) == 0:
            raise RuntimeError('No overlaping weights between model file and pretrained weight file. Please check')
        model_dict.update(overlap)
        model.load_state_dict(model_dict)
    return model
