"""nn_layers/aspp.py on the HIP path: `ASPP(num_classes)` (512-channel trunks) and `ASPP_Bottleneck(num_classes)` (2048-channel
trunks) -- the DeepLabv3 heads of BASELINE configs[4].  Same constructor, attribute names and `state_dict()` as the reference;
eval-mode BatchNorm only (folded, together with the conv bias, into the producing kernel's epilogue).

    out_1x1   = relu(bn(conv_1x1_1(x)))                       aspp.py:40 / :87
    out_3x3_k = relu(bn(conv_3x3_k(x))),  dilation 6, 12, 18  aspp.py:41-43 / :88-90      <- K13, the matrix-core kernel
    out_img   = interpolate(relu(bn(conv_1x1_2(avg_pool(x)))))  (a constant map: bilinear from 1x1)   :45-47 / :92-94
    out       = conv_1x1_4(relu(bn(conv_1x1_3(cat[...]))))     aspp.py:49-51 / :96-98

The five branches write straight into their channel slice of one (N,1280,H,W) buffer.
"""
import torch
from torch import nn

from . import ops
from .layers import bn_fold, cached
from .ops import Epi


class _ASPPBase(nn.Module):
    def __init__(self, in_channels, num_classes):
        super().__init__()
        self.conv_1x1_1 = nn.Conv2d(in_channels, 256, kernel_size=1)
        self.bn_conv_1x1_1 = nn.BatchNorm2d(256)
        self.conv_3x3_1 = nn.Conv2d(in_channels, 256, kernel_size=3, stride=1, padding=6, dilation=6)
        self.bn_conv_3x3_1 = nn.BatchNorm2d(256)
        self.conv_3x3_2 = nn.Conv2d(in_channels, 256, kernel_size=3, stride=1, padding=12, dilation=12)
        self.bn_conv_3x3_2 = nn.BatchNorm2d(256)
        self.conv_3x3_3 = nn.Conv2d(in_channels, 256, kernel_size=3, stride=1, padding=18, dilation=18)
        self.bn_conv_3x3_3 = nn.BatchNorm2d(256)
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_1x1_2 = nn.Conv2d(in_channels, 256, kernel_size=1)
        self.bn_conv_1x1_2 = nn.BatchNorm2d(256)
        self.conv_1x1_3 = nn.Conv2d(1280, 256, kernel_size=1)       # (1280 = 5*256)
        self.bn_conv_1x1_3 = nn.BatchNorm2d(256)
        self.conv_1x1_4 = nn.Conv2d(256, num_classes, kernel_size=1)

    def _packed(self, conv):
        return cached(conv, 'packed', [conv.weight], lambda: ops.pack_dense_weight(conv.weight))

    def _epi(self, conv, bn, ctot=None, coff=0):
        """relu(bn(conv(x) + bias)) as scale / shift / alpha = 0: shift = bn_shift + bias * bn_scale.  Epilogue vectors are
        indexed by the DESTINATION channel: for a branch written into its slice of the concatenation they are laid out at
        that offset in a ctot-sized vector."""
        def build():
            scale, shift = bn_fold(bn)
            shift = shift + conv.bias * scale
            n = scale.numel()
            tot = n if ctot is None else ctot
            sc, sh = torch.ones(tot, device=scale.device), torch.zeros(tot, device=scale.device)
            sc[coff:coff + n] = scale
            sh[coff:coff + n] = shift
            return sc, sh, torch.zeros(tot, device=scale.device)
        sc, sh, zero = cached(bn, 'aspp_epi_%s_%d' % (ctot, coff), [bn.weight, bn.bias, bn.running_mean, bn.running_var, conv.bias], build)
        return Epi(sc, sh, zero)

    def forward(self, feature_map):
        if torch.is_grad_enabled():
            raise RuntimeError('mspl_amd: the ASPP heads are inference-only on the HIP path (call under torch.no_grad())')
        N, _, H, W = feature_map.shape
        cat = torch.empty((N, 1280, H, W), device=feature_map.device, dtype=torch.float32)
        ops.dense_conv(feature_map, self._packed(self.conv_1x1_1), 1, 1, self._epi(self.conv_1x1_1, self.bn_conv_1x1_1, 1280, 0),
                       out=(cat, 0))
        for i, (conv, bn) in enumerate([(self.conv_3x3_1, self.bn_conv_3x3_1), (self.conv_3x3_2, self.bn_conv_3x3_2),
                                        (self.conv_3x3_3, self.bn_conv_3x3_3)]):
            ops.dense_conv(feature_map, self._packed(conv), 3, conv.dilation[0], self._epi(conv, bn, 1280, 256 * (i + 1)),
                           out=(cat, 256 * (i + 1)))
        img = ops.adaptive_avgpool(feature_map, (1, 1))
        img = ops.dense_conv(img, self._packed(self.conv_1x1_2), 1, 1, self._epi(self.conv_1x1_2, self.bn_conv_1x1_2))
        ops.bilinear(img, (H, W), out=(cat, 1024))            # bilinear from a 1x1 map: the constant, whatever align_corners is
        out = ops.dense_conv(cat, self._packed(self.conv_1x1_3), 1, 1, self._epi(self.conv_1x1_3, self.bn_conv_1x1_3))
        c4 = self.conv_1x1_4
        return ops.conv1x1(out, c4.weight, 1, Epi(shift=c4.bias))


def upsample_logits(logits, size):
    """The head's caller: F.upsample(output, size=(h, w), mode='bilinear') with the default align_corners=False
    (model/segmentation/deeplabv3.py:40)."""
    return ops.bilinear(logits, size, align_corners=False)


class ASPP(_ASPPBase):
    def __init__(self, num_classes):
        super().__init__(512, num_classes)


class ASPP_Bottleneck(_ASPPBase):
    def __init__(self, num_classes):
        super().__init__(4 * 512, num_classes)
