#!/usr/bin/env python3
"""Algorithmic bytes of the uest train step per layer kind, in SURVEY.md 8(d)'s accounting: a convolution moves
its input and its output once (the four EESP branches share one read; BatchNorm / PReLU / add / cat / Shuffle fused = 0), a resample
its input and output.  Forward (F), data gradient (D: reads dL/dout, writes dL/din = the same two tensors) and weight gradient (W:
reads dL/dout and the input = the same two tensors) of a weighted layer therefore move 3x its forward bytes; a weightless layer 2x.
ESPDNet-UE s=2.0, C classes, H x W input, per image, fp32.   usage: python tools/train_bytes.py [C H W]"""
import math, sys
C, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (5, 256, 480)
rows = {}
BN = [0.0]        # bytes of the batch-statistics passes (the supervised loop, model.train()): per BatchNorm one read of its input for the
                  # statistics (forward) and one pass over (gradient, input) for the two channel sums (backward): 3 x 4 * C * HW


def add(kind, fwd_bytes, weighted=True):
    r = rows.setdefault(kind, [0.0, 0.0])
    r[0] += fwd_bytes
    r[1] += fwd_bytes * (3 if weighted else 2)


def px(div):
    return (H // div) * (W // div)


def bn(channels, pixels):
    BN[0] += 3 * 4 * channels * pixels


def eesp(kind, cin, cout, div_in, stride):
    n = cout // 4
    pi, po = px(div_in), px(div_in * stride)
    bn(n, pi); bn(4 * n, po); bn(cout, po)            # proj_1x1.bn, br_after_cat, conv_1x1_exp.bn
    add(kind + ' proj_1x1 (grouped 1x1)', 4 * pi * (cin + n))
    add(kind + ' K2 (4 dilated depthwise 3x3 + HFF)', 4 * n * (pi + 4 * po))
    add(kind + ' conv_1x1_exp (grouped 1x1)', 4 * po * (cout + cout))


def down(kind, cin, cout, div_in):
    eesp(kind, cin, cout - cin, div_in, 2)
    add(kind + ' avg pool 3x3/s2', 4 * cin * (px(div_in) + px(div_in * 2)), weighted=False)
    add(kind + ' inp_reinf (3x3 on the pooled image + 1x1)', 4 * px(div_in * 2) * (3 + 3 + 3 + cout))
    bn(3, px(div_in * 2)); bn(cout, px(div_in * 2))   # inp_reinf's two BatchNorms


def pyr(kind, cin, cout, div):
    p = px(div)
    h, w = H // div, W // div
    add(kind + ' projection 1x1', 4 * p * (cin + 16))
    bn(16, p); bn(80, p); bn(16, p)                   # projection_layer, merge_layer.0, merge_layer.2
    if cout != C:
        bn(cout, p)                                   # last_layer_br (not on the two classifier pyramids)
    for s in (2.0, 1.5, 1.0, 0.5, 0.1):
        hs, ws = max(math.ceil(h * s), 5), max(math.ceil(w * s), 5)
        # resample in, depthwise 3x3 at the branch resolution, resample out (node-per-op accounting of SURVEY 8d)
        if s == 1.0:
            add(kind + ' branches (resample + depthwise 3x3 + resample)', 4 * 16 * 2 * p)
        else:
            add(kind + ' branches (resample + depthwise 3x3 + resample)', 4 * 16 * (p + hs * ws), weighted=False)
            add(kind + ' branches (resample + depthwise 3x3 + resample)', 4 * 16 * 2 * hs * ws)
            add(kind + ' branches (resample + depthwise 3x3 + resample)', 4 * 16 * (hs * ws + p), weighted=False)
    add(kind + ' merge (grouped 3x3 80 -> 16) + final 1x1', 4 * p * (80 + 16) + 4 * p * (16 + cout))


add('level1 CBR 3x3/s2', 4 * (3 * px(1) + 32 * px(2)))
bn(32, px(2))
down('level2_0', 32, 128, 2)
down('level3_0', 128, 256, 4)
for _ in range(3):
    eesp('level3 x3', 256, 256, 8, 1)
down('level4_0', 256, 512, 8)
for _ in range(7):
    eesp('level4 x7', 512, 512, 16, 1)
pyr('bu_dec_l1', 512, 64, 16)
pyr('bu_dec_l2', 64, 48, 8)
pyr('bu_dec_l3', 48, 32, 4)
pyr('aux_decoder', 32, C, 4)
pyr('bu_dec_l4', 32, C, 2)
for name, cin, cout, div in (('merge_enc_dec_l2', 256, 64, 8), ('merge_enc_dec_l3', 128, 48, 4), ('merge_enc_dec_l4', 32, 32, 2)):
    add('EfficientPWConv x3 (grouped 3x3 + gate)', 4 * px(div) * (cin + cout) + 4 * px(div) * cin)
    add('decoder up-merge x3 (bilinear x2 + add + BR)', 4 * cout * (px(div * 2) + 2 * px(div)), weighted=False)
    bn(cout, px(div)); bn(cout, px(div)); bn(cout, px(div))          # EfficientPWConv's two BatchNorms, bu_br
add('final bilinear of both heads + loss', 4 * C * (px(2) + px(4)) + 2 * 4 * C * px(1), weighted=False)
tf = sum(r[0] for r in rows.values())
tt = sum(r[1] for r in rows.values())
groups = {}
for k, r in rows.items():
    g = k.split(' ')[0] if not k.startswith(('bu_dec', 'aux')) else 'pyramids'
    g = {'level2_0': 'DownSamplers', 'level3_0': 'DownSamplers', 'level4_0': 'DownSamplers', 'level3': 'EESP stride 1 (x10)', 'level4': 'EESP stride 1 (x10)',
         'level1': 'stem', 'EfficientPWConv': 'skip connections + up-merges', 'decoder': 'skip connections + up-merges', 'final': 'heads + loss'}.get(g, g)
    a = groups.setdefault(g, [0.0, 0.0])
    a[0] += r[0]
    a[1] += r[1]
print('| layer group | forward MB / image | forward + backward MB / image |')
print('|---|---|---|')
for g, (f, t) in groups.items():
    print('| %s | %.1f | %.1f |' % (g, f / 1e6, t / 1e6))
print('| **total** | **%.1f** | **%.1f** |' % (tf / 1e6, tt / 1e6))
print('batch-statistics passes of the BatchNorms in train() (statistics read + backward sums pass): %.1f MB / image -> %.1f MB / image with them'
      % (BN[0] / 1e6, (tt + BN[0]) / 1e6))
