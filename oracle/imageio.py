"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the image/label I/O on either side of the hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

What is restated (reference file:line):
  data_loader/segmentation/greenhouse.py:207-225,232-262   val_transforms = Resize(size) -> Normalize(); PIL decode
  transforms/segmentation/data_transforms.py:191-212       Resize: rgb/depth PIL BILINEAR, label PIL NEAREST
  transforms/segmentation/data_transforms.py:15-46         Tensorize / Normalize: to_tensor (/255), normalize((x-MEAN)/STD)
  transforms/classification/data_transforms.py:10-11       MEAN, STD
  uest_seg_multi_os.py:720-728                             update_image_list
  uest_seg_multi_os.py:923-931                             label PNG (8-bit, single channel) via PIL

The resampling itself lives in third-party dependencies that are NOT part of /root/reference:
  * Pillow (requirements.txt pins nothing; 12.2.0 is what this image has): `Image.resize` -> libImaging/Resample.c
    (`precompute_coeffs`, `normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc`, `ImagingResampleVertical_8bpc`:
    triangle filter whose support grows with the down-scale factor, 22-bit fixed-point coefficients, horizontal pass
    first, uint8 rounding between the passes) and libImaging/Geometry.c (`ImagingScaleAffine`, NEAREST).
  * torchvision (>=0.3.0, absent here): `functional.to_tensor` = HWC uint8 -> CHW float32 / 255;
    `functional.normalize` = (t - mean) / std in float32.
Their published algorithms are restated below; the restatement is PINNED against Pillow itself (importable here) by
tests/golden/make_golden.py::gen_imageio (vectors in tests/golden/imageio.npz) and live in tests/test_imageio.py.
"""
import math
import zlib
import struct

import numpy as np

MEAN = (0.485, 0.456, 0.406)      # transforms/classification/data_transforms.py:10
STD = (0.229, 0.224, 0.225)       # :11
PRECISION_BITS = 32 - 8 - 2       # Resample.c


def _bilinear_filter(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full box (0, in_size), BILINEAR (support 1.0).
    Returns (bounds int32 (out,2) = (first, count), coefficients int32 (out, ksize))."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bilinear_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis0(img, out_size):
    """One pass along axis 0 of a uint8 array (any trailing shape)."""
    bounds, kk = precompute_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for i in range(out_size):
        a, n = bounds[i]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for j in range(n):
            acc += src[a + j] * int(kk[i, j])
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bilinear_u8(img, size):
    """PIL `Image.resize(size, Image.BILINEAR)` on an 8-bit image.  img: (H,W) or (H,W,C) uint8; size = (W_out, H_out)
    (PIL order).  Horizontal pass first; a pass whose size does not change is skipped (Resample.c ImagingResampleInner)."""
    w_out, h_out = size
    img = np.ascontiguousarray(img)
    if img.shape[1] != w_out:
        img = np.swapaxes(_resample_axis0(np.swapaxes(img, 0, 1), w_out), 0, 1)
    if img.shape[0] != h_out:
        img = _resample_axis0(img, h_out)
    return np.ascontiguousarray(img)


def nearest_index(in_size, out_size):
    """Source index per destination index for PIL NEAREST (Geometry.c ImagingScaleAffine: xo starts at a[2] + a[0]*0.5 and
    is advanced by a running `xo += a[0]` in double precision; index = (int)xo when 0 <= xo < in_size)."""
    a0 = float(in_size) / out_size
    idx = np.empty(out_size, np.int64)
    xo = a0 * 0.5
    for x in range(out_size):
        idx[x] = min(int(xo), in_size - 1) if xo >= 0 else 0
        xo += a0
    return idx


def resize_nearest_u8(img, size):
    """PIL `Image.resize(size, Image.NEAREST)` (label maps, transforms/segmentation/data_transforms.py:203)."""
    w_out, h_out = size
    if img.shape[0] == h_out and img.shape[1] == w_out:
        return img.copy()
    return img[nearest_index(img.shape[0], h_out)][:, nearest_index(img.shape[1], w_out)]


def to_tensor(img):
    """torchvision.transforms.functional.to_tensor for uint8 HWC / HW arrays: CHW float32 in [0,1]."""
    if img.ndim == 2:
        img = img[:, :, None]
    return (img.transpose(2, 0, 1).astype(np.float32) / np.float32(255))


def normalize(t, mean=MEAN, std=STD):
    """torchvision.transforms.functional.normalize: (t - mean) / std per channel, float32."""
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((t - m) / s).astype(np.float32)


def val_transform(rgb, label=None, depth=None, size=(480, 256), normalise=True, flip=False):
    """greenhouse.py:216-222 val_transforms (Resize -> Normalize | Tensorize) (+ RandomFlip's mirror when flip=True, as the
    train transforms apply it between the two, data_transforms.py:49-66).  rgb (H,W,3) uint8, label (H,W) uint8,
    depth (H,W) uint8.  Returns (rgb float32 (3,h,w), label int64 (h,w) | None, depth float32 (1,h,w) | None)."""
    rgb = resize_bilinear_u8(rgb, size)
    label = None if label is None else resize_nearest_u8(label, size)
    depth = None if depth is None else resize_bilinear_u8(depth, size)
    if flip:
        rgb = rgb[:, ::-1]
        label = None if label is None else label[:, ::-1]
        depth = None if depth is None else depth[:, ::-1]
    t = to_tensor(rgb)
    if normalise:
        t = normalize(t)
    return (t, None if label is None else label.astype(np.int64), None if depth is None else to_tensor(depth))


def png_decode_gray8(data):
    """Minimal PNG reader for 8-bit single-channel non-interlaced files (what uest_seg_multi_os.py:929-931 writes):
    used to check the product's writer without PIL."""
    assert data[:8] == b'\x89PNG\r\n\x1a\n'
    pos, idat, w = 8, b'', None
    while pos < len(data):
        n, typ = struct.unpack('>I4s', data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack('>I', data[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(typ + body) & 0xffffffff)
        if typ == b'IHDR':
            w, h, depth, ctype, comp, flt, inter = struct.unpack('>IIBBBBB', body)
            assert (depth, ctype, comp, flt, inter) == (8, 0, 0, 0, 0)
        elif typ == b'IDAT':
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w + 1)
    out = np.zeros((h, w), np.uint8)
    for y in range(h):
        f, row = raw[y, 0], raw[y, 1:].astype(np.int32)
        up = out[y - 1].astype(np.int32) if y else np.zeros(w, np.int32)
        if f == 0:
            out[y] = row
        elif f == 2:
            out[y] = (row + up) & 255
        elif f == 1:
            acc = 0
            for x in range(w):
                acc = (row[x] + acc) & 255
                out[y, x] = acc
        elif f in (3, 4):
            left = ul = 0
            for x in range(w):
                if f == 3:
                    pred = (left + up[x]) >> 1
                else:
                    p = left + up[x] - ul
                    pa, pb, pc = abs(p - left), abs(p - up[x]), abs(p - ul)
                    pred = left if (pa <= pb and pa <= pc) else (up[x] if pb <= pc else ul)
                left = (row[x] + pred) & 255
                ul = up[x]
                out[y, x] = left
        else:
            raise ValueError('bad PNG filter %d' % f)
    return out
