"""Oracle: functional CPU restatement of the ESPNetv2 / ESPDNet-UE forward.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain torch-CPU fp32
functional ops driven directly by a reference-format ``state_dict`` -- no
nn.Module tree -- so it is an independent statement of the algorithm the HIP
path must reproduce.  Every function cites the reference lines it follows
(paths relative to /root/reference).

Pinned by tests/test_oracle_golden.py against vectors generated from the real
reference modules (tests/golden/make_golden.py).
"""
import math

import torch
import torch.nn.functional as F

# model/classification/espnetv2_config.py:6-23
SC_CH = {
    0.5: [16, 32, 64, 128, 256, 1024],
    1.0: [32, 64, 128, 256, 512, 1024],
    1.25: [32, 80, 160, 320, 640, 1024],
    1.5: [32, 96, 192, 384, 768, 1024],
    2.0: [32, 128, 256, 512, 1024, 1280],
}
REP_LAYERS = [0, 3, 7, 3]
RECEPT_LIMIT = [13, 11, 9, 7, 5]
BRANCHES = 4
PYR_SCALES = [2.0, 1.5, 1.0, 0.5, 0.1]  # nn_layers/efficient_pyramid_pool.py:15,18 (sorted descending)
BN_EPS = 1e-5


def eesp_dilations(r_lim, k=BRANCHES):
    """nn_layers/eesp.py:38-53: kernel sizes 3,5,7,.. capped at r_lim -> 3, sorted, mapped to dilation."""
    ks = []
    for i in range(k):
        ksize = 3 + 2 * i
        ks.append(ksize if ksize <= r_lim else 3)
    ks.sort()
    return [(ksz - 1) // 2 for ksz in ks]


# ---------------------------------------------------------------- primitives
BN_MODE = {'training': False, 'momentum': 0.1}


class bn_training(object):
    """Context: BatchNorm uses batch statistics and updates sd[...running_mean/var] in place (nn.BatchNorm2d in train(),
    the supervised loop's model.train(), utilities/train_eval_seg.py:174)."""

    def __enter__(self):
        BN_MODE['training'] = True

    def __exit__(self, *exc):
        BN_MODE['training'] = False


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        BN_MODE['training'], BN_MODE['momentum'], BN_EPS)


def _enc_conv(x, sd, p, stride=1, groups=1, dilation=1):
    """nn_layers/espnet_utils.py:23,78,107,131: bias-free conv, pad = (k-1)/2 * d."""
    w = sd[p + '.conv.weight']
    pad = ((w.shape[-1] - 1) // 2) * dilation
    return F.conv2d(x, w, None, stride, pad, dilation, groups)


def enc_cbr(x, sd, p, stride=1, groups=1):
    """espnet_utils.CBR, nn_layers/espnet_utils.py:8-37."""
    return F.prelu(_bn(_enc_conv(x, sd, p, stride, groups), sd, p + '.bn'), sd[p + '.act.weight'])


def enc_cb(x, sd, p, stride=1, groups=1):
    """espnet_utils.CB, nn_layers/espnet_utils.py:62-89."""
    return _bn(_enc_conv(x, sd, p, stride, groups), sd, p + '.bn')


def enc_br(x, sd, p):
    """espnet_utils.BR, nn_layers/espnet_utils.py:39-60."""
    return F.prelu(_bn(x, sd, p + '.bn'), sd[p + '.act.weight'])


def dec_cbr(x, sd, p, groups=1):
    """cnn_utils.CBR (Sequential keys cbr.0/1/2), nn_layers/cnn_utils.py:26-54."""
    w = sd[p + '.cbr.0.weight']
    pad = (w.shape[-1] - 1) // 2
    y = F.conv2d(x, w, None, 1, pad, 1, groups)
    return F.prelu(_bn(y, sd, p + '.cbr.1'), sd[p + '.cbr.2.weight'])


def dec_br(x, sd, p):
    """cnn_utils.BR (keys br.0/1), nn_layers/cnn_utils.py:85-105."""
    return F.prelu(_bn(x, sd, p + '.br.0'), sd[p + '.br.1.weight'])


def seq_br(x, sd, p):
    """bare nn.Sequential(BatchNorm2d, PReLU) (keys 0/1), model/segmentation/espdnet_ue.py:89-97."""
    return F.prelu(_bn(x, sd, p + '.0'), sd[p + '.1.weight'])


def shuffle(x, groups):
    """nn_layers/cnn_utils.py:119-125."""
    n, c, h, w = x.shape
    return x.view(n, groups, c // groups, h, w).transpose(1, 2).contiguous().view(n, c, h, w)


# ---------------------------------------------------------------- blocks
def eesp(x, sd, p, stride, r_lim, down_avg=False, k=BRANCHES):
    """nn_layers/eesp.py:60-93: reduce -> split/transform (dilated depthwise) -> HFF -> merge."""
    dil = eesp_dilations(r_lim, k)
    o1 = enc_cbr(x, sd, p + '.proj_1x1', 1, k)
    n = o1.shape[1]
    outs = []
    for i in range(k):
        w = sd['%s.spp_dw.%d.conv.weight' % (p, i)]
        o = F.conv2d(o1, w, None, stride, dil[i], dil[i], n)
        if i > 0:
            o = o + outs[i - 1]
        outs.append(o)
    cat = torch.cat(outs, 1)
    expanded = enc_cb(enc_br(cat, sd, p + '.br_after_cat'), sd, p + '.conv_1x1_exp', 1, k)
    if stride == 2 and down_avg:
        return expanded
    if expanded.shape == x.shape:
        expanded = expanded + x
    return F.prelu(expanded, sd[p + '.module_act.weight'])


def downsampler(x, sd, p, r_lim, image=None, k=BRANCHES):
    """nn_layers/eesp.py:123-144."""
    avg_out = F.avg_pool2d(x, 3, 2, 1)
    eesp_out = eesp(x, sd, p + '.eesp', 2, r_lim, down_avg=True, k=k)
    out = torch.cat([avg_out, eesp_out], 1)
    if image is not None:
        h1 = avg_out.shape[2]
        while True:  # height-only match, eesp.py:135-140
            image = F.avg_pool2d(image, 3, 2, 1)
            if image.shape[2] == h1:
                break
        r = enc_cbr(image, sd, p + '.inp_reinf.0')
        out = out + enc_cb(r, sd, p + '.inp_reinf.1')
    return F.prelu(out, sd[p + '.act.weight'])


def pyr_pool(x, sd, p, last_layer_br=True, scales=PYR_SCALES):
    """nn_layers/efficient_pyramid_pool.py:36-61."""
    x = dec_cbr(x, sd, p + '.projection_layer')
    h, w = x.shape[2:]
    c = x.shape[1]
    hs = []
    for i, s in enumerate(scales):
        wt = sd['%s.stages.%d.weight' % (p, i)]
        h_s = max(int(math.ceil(h * s)), 5)
        w_s = max(int(math.ceil(w * s)), 5)
        if s < 1.0:
            t = F.adaptive_avg_pool2d(x, (h_s, w_s))
            t = F.conv2d(t, wt, None, 1, 1, 1, c)
            t = F.interpolate(t, (h, w), mode='bilinear', align_corners=True)
        elif s > 1.0:
            t = F.interpolate(x, (h_s, w_s), mode='bilinear', align_corners=True)
            t = F.conv2d(t, wt, None, 1, 1, 1, c)
            t = F.adaptive_avg_pool2d(t, (h, w))
        else:
            t = F.conv2d(x, wt, None, 1, 1, 1, c)
        hs.append(t)
    out = torch.cat(hs, 1)
    out = dec_br(out, sd, p + '.merge_layer.0')
    out = shuffle(out, len(scales))
    out = dec_cbr(out, sd, p + '.merge_layer.2', groups=c)
    bias = sd.get(p + '.merge_layer.3.bias', None)
    out = F.conv2d(out, sd[p + '.merge_layer.3.weight'], bias)
    if last_layer_br:
        out = dec_br(out, sd, p + '.br')
    return out


def pw_conv(x, sd, p):
    """nn_layers/efficient_pt.py:25-29."""
    wt = torch.sigmoid(F.conv2d(F.adaptive_avg_pool2d(x, 1), sd[p + '.wt_layer.1.weight']))
    w = sd[p + '.expansion_layer.cbr.0.weight']
    groups = math.gcd(x.shape[1], w.shape[0])
    return dec_cbr(x, sd, p + '.expansion_layer', groups=groups) * wt


def up2(x):
    return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)


# ---------------------------------------------------------------- encoders
def _encoder(x, sd, image_for_l2, l3_tail_prefix):
    """model/classification/espnetv2.py:61-72 topology as called by the segmentation nets.

    image_for_l2: the image handed to level2_0 (ESPDNet-UE passes x, espdnet_ue.py:197; ESPNetv2
    passes nothing, espnetv2.py:127).  l3_tail_prefix: module prefix used for level3[1:], which is
    'depth_base_net' in ESPDNet-UE (espdnet_ue.py:226) and 'base_net' in ESPNetv2 (espnetv2.py:130-134).
    """
    b = 'base_net'
    l1 = enc_cbr(x, sd, b + '.level1', stride=2)
    l2 = downsampler(l1, sd, b + '.level2_0', RECEPT_LIMIT[0], image_for_l2)
    l3 = downsampler(l2, sd, b + '.level3_0', RECEPT_LIMIT[1], x)
    for i in range(REP_LAYERS[1]):
        pre = b if i == 0 else l3_tail_prefix
        l3 = eesp(l3, sd, '%s.level3.%d' % (pre, i), 1, RECEPT_LIMIT[2])
    l4 = downsampler(l3, sd, b + '.level4_0', RECEPT_LIMIT[2], x)
    for i in range(REP_LAYERS[2]):
        l4 = eesp(l4, sd, '%s.level4.%d' % (b, i), 1, RECEPT_LIMIT[3])
    return l1, l2, l3, l4


def fusion_gate(rgb, depth, sd, p, trainable=True):
    """nn_layers/fusion_gate.py:26-47 (the `.to('cuda')` of :38 dropped: everything stays on the inputs' device)."""
    if not trainable:
        return rgb + depth
    w = torch.sigmoid(F.conv2d(torch.cat((rgb, depth), 1), sd[p + '.conv_1x1.conv.weight']))
    return rgb * w + depth * (torch.ones_like(w) - w)


def _encoder_rgbd(x, x_d, sd, dense_fuse, trainable_fusion):
    """espdnet_ue.py:186-270 with x_d given: the depth branch's DownSamplers get no image reinforcement
    (:198,:209,:240); the RGB branch's level3[1:] still run through depth_base_net's blocks (:226)."""
    b, d = 'base_net', 'depth_base_net'
    gate = lambda r, dd, lvl: fusion_gate(r, dd, sd, 'fusion_gate_level%d' % lvl, trainable_fusion)
    dl1 = enc_cbr(x_d, sd, d + '.level1', stride=2)
    l1 = gate(enc_cbr(x, sd, b + '.level1', stride=2), dl1, 1)
    dl2 = downsampler(dl1, sd, d + '.level2_0', RECEPT_LIMIT[0], None)
    l2 = gate(downsampler(l1, sd, b + '.level2_0', RECEPT_LIMIT[0], x), dl2, 2)
    l3 = downsampler(l2, sd, b + '.level3_0', RECEPT_LIMIT[1], x)
    dl3 = downsampler(dl2, sd, d + '.level3_0', RECEPT_LIMIT[1], None)
    for i in range(REP_LAYERS[1]):
        l3 = eesp(l3, sd, '%s.level3.%d' % (b if i == 0 else d, i), 1, RECEPT_LIMIT[2])
        dl3 = eesp(dl3, sd, '%s.level3.%d' % (d, i), 1, RECEPT_LIMIT[2])
        if dense_fuse:
            l3 = gate(l3, dl3, 3)
    if not dense_fuse:
        l3 = gate(l3, dl3, 3)
    l4 = downsampler(l3, sd, b + '.level4_0', RECEPT_LIMIT[2], x)
    dl4 = downsampler(dl3, sd, d + '.level4_0', RECEPT_LIMIT[2], None)
    for i in range(REP_LAYERS[2]):
        l4 = eesp(l4, sd, '%s.level4.%d' % (b, i), 1, RECEPT_LIMIT[3])
        dl4 = eesp(dl4, sd, '%s.level4.%d' % (d, i), 1, RECEPT_LIMIT[3])
        if dense_fuse:
            l4 = gate(l4, dl4, 4)
    if not dense_fuse:
        l4 = gate(l4, dl4, 4)
    return l1, l2, l3, l4


def _decoder(sd, l1, l2, l3, l4, with_aux):
    """model/segmentation/espdnet_ue.py:272-299 / espnetv2.py:144-165."""
    bu = pyr_pool(l4, sd, 'bu_dec_l1')
    bu = seq_br(pw_conv(l3, sd, 'merge_enc_dec_l2') + up2(bu), sd, 'bu_br_l2')
    bu = pyr_pool(bu, sd, 'bu_dec_l2')
    bu = seq_br(pw_conv(l2, sd, 'merge_enc_dec_l3') + up2(bu), sd, 'bu_br_l3')
    bu = pyr_pool(bu, sd, 'bu_dec_l3')
    aux = pyr_pool(bu, sd, 'aux_decoder', last_layer_br=False) if with_aux else None
    bu = seq_br(pw_conv(l1, sd, 'merge_enc_dec_l4') + up2(bu), sd, 'bu_br_l4')
    bu = pyr_pool(bu, sd, 'bu_dec_l4', last_layer_br=False)
    return bu, aux


def espdnet_ue_forward(sd, x, x_d=None, dense_fuse=False, trainable_fusion=True):
    """ESPDNetwithUncertaintyEstimation.forward, model/segmentation/espdnet_ue.py:169-302 (x_d: the RGB-D path).

    Returns (main, aux) logits at input resolution.  Raises RuntimeError (from the tensor add) when
    H or W is not a multiple of 16, exactly like the reference.
    """
    size = x.shape[2:]
    if x_d is None:
        l1, l2, l3, l4 = _encoder(x, sd, x, 'depth_base_net')
    else:
        l1, l2, l3, l4 = _encoder_rgbd(x, x_d, sd, dense_fuse, trainable_fusion)
    bu, aux = _decoder(sd, l1, l2, l3, l4, True)
    return (F.interpolate(bu, size=size, mode='bilinear', align_corners=True),
            F.interpolate(aux, size=size, mode='bilinear', align_corners=True))


def espdnet_forward(sd, x, x_d=None, dense_fuse=False, trainable_fusion=True):
    """ESPDNetSegmentation.forward, model/segmentation/espdnet.py:183-309: ESPDNet-UE without the auxiliary head."""
    size = x.shape[2:]
    if x_d is None:
        l1, l2, l3, l4 = _encoder(x, sd, x, 'depth_base_net')
    else:
        l1, l2, l3, l4 = _encoder_rgbd(x, x_d, sd, dense_fuse, trainable_fusion)
    bu, _ = _decoder(sd, l1, l2, l3, l4, False)
    return F.interpolate(bu, size=size, mode='bilinear', align_corners=True)


def espnetv2_forward(sd, x):
    """ESPNetv2Segmentation.forward, model/segmentation/espnetv2.py:115-167."""
    size = x.shape[2:]
    l1, l2, l3, l4 = _encoder(x, sd, None, 'base_net')
    bu, _ = _decoder(sd, l1, l2, l3, l4, False)
    return F.interpolate(bu, size=size, mode='bilinear', align_corners=True)


def aspp_forward(sd, x, p=''):
    """ASPP.forward / ASPP_Bottleneck.forward, nn_layers/aspp.py:34-52 / :80-99 (the two differ in input width only).

    Biased convolutions + eval-mode BatchNorm + ReLU; the image-pooling branch is interpolated with the DEFAULT
    align_corners=False (from a 1x1 map, i.e. a constant).  sd: reference-format state dict, p: key prefix ('' or 'head.').
    """
    def k(name):
        return sd[p + name]

    def cbr(t, conv, bn, **kw):
        t = F.conv2d(t, k(conv + '.weight'), k(conv + '.bias'), **kw)
        t = F.batch_norm(t, k(bn + '.running_mean'), k(bn + '.running_var'), k(bn + '.weight'), k(bn + '.bias'), False, 0.0, 1e-5)
        return F.relu(t)
    h, w = x.shape[2:]
    outs = [cbr(x, 'conv_1x1_1', 'bn_conv_1x1_1')]
    for i, d in enumerate((6, 12, 18)):
        outs.append(cbr(x, 'conv_3x3_%d' % (i + 1), 'bn_conv_3x3_%d' % (i + 1), padding=d, dilation=d))
    img = cbr(F.adaptive_avg_pool2d(x, 1), 'conv_1x1_2', 'bn_conv_1x1_2')
    outs.append(F.interpolate(img, size=(h, w), mode='bilinear'))
    out = cbr(torch.cat(outs, 1), 'conv_1x1_3', 'bn_conv_1x1_3')
    return F.conv2d(out, k('conv_1x1_4.weight'), k('conv_1x1_4.bias'))
