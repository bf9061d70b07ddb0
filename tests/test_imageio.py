"""Loader transforms, label PNG writer and list round trip.

CPU: the oracle's restatement of Pillow's resampling / torchvision's to_tensor+normalize against vectors produced by Pillow
itself (tests/golden/make_golden.py::gen_imageio), the PNG writer against PIL's decoder and the oracle's own decoder, the list
file round trip.  GPU: the batched device transforms (through the C ABI) bit-exact against the same vectors and the oracle;
the asynchronous writer end to end.
"""
import hashlib
import io
import os

import numpy as np
import pytest
import torch

from oracle import imageio as oio
from tests.cases import IMAGEIO_CASES
from tests.synth import synth_image_u8


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


@pytest.mark.parametrize('name', sorted(IMAGEIO_CASES))
def test_oracle_transforms_vs_pillow_golden(name, golden):
    hs, ws, size, seed, norm, flip, with_depth = IMAGEIO_CASES[name]
    rgb, label, depth = synth_image_u8(hs, ws, seed)
    t, lt, dt = oio.val_transform(rgb, label, depth if with_depth else None, size=size, normalise=norm, flip=flip)
    g = golden('imageio')
    assert np.array_equal(t[:, ::7, ::5], g[name + '.rgb_s'])
    assert np.array_equal(_sha(t), g[name + '.rgb_sha'])
    assert np.array_equal(_sha(lt), g[name + '.label_sha'])
    if with_depth:
        assert np.array_equal(_sha(dt), g[name + '.depth_sha'])


def test_oracle_resample_vs_live_pillow():
    """Extra sizes, checked against the Pillow installed here (skipped where PIL is missing)."""
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.default_rng(3)
    for hs, ws, size in [(61, 47, (31, 90)), (300, 400, (133, 77)), (90, 160, (160, 90)), (256, 480, (480, 256))]:
        img = rng.integers(0, 256, (hs, ws, 3), dtype=np.uint8)
        lab = rng.integers(0, 6, (hs, ws), dtype=np.uint8)
        assert np.array_equal(oio.resize_bilinear_u8(img, size), np.asarray(Image.fromarray(img).resize(size, Image.BILINEAR)))
        assert np.array_equal(oio.resize_nearest_u8(lab, size), np.asarray(Image.fromarray(lab).resize(size, Image.NEAREST)))


def test_host_tables_match_oracle():
    """The C ABI's host-side coefficient builders (the only part of the transforms that runs without a GPU)."""
    from mspl_amd._native import check, lib
    for n_in, n_out in [(360, 256), (360, 288), (720, 256), (100, 256), (37, 48), (53, 64), (256, 256), (2048, 1024), (5, 7)]:
        k = lib.mspl_resample_ksize(n_in, n_out)
        bounds, kk = np.zeros((n_out, 2), np.int32), np.zeros((n_out, k), np.int32)
        check(lib.mspl_resample_coeffs(n_in, n_out, bounds.ctypes.data, kk.ctypes.data))
        ob, ok = oio.precompute_coeffs(n_in, n_out)
        assert k == ok.shape[1] and np.array_equal(bounds, ob) and np.array_equal(kk, ok), (n_in, n_out)
        idx = np.zeros(n_out, np.int32)
        check(lib.mspl_nearest_index(n_in, n_out, idx.ctypes.data))
        assert np.array_equal(idx, oio.nearest_index(n_in, n_out)), (n_in, n_out)
    assert lib.mspl_resample_ksize(0, 4) < 0


def test_png_writer_round_trip():
    from mspl_amd.io import encode_png_gray8
    rng = np.random.default_rng(7)
    cases = [np.repeat(np.repeat(rng.integers(0, 5, (16, 30), dtype=np.uint8), 16, 0), 16, 1),      # blocky label map
             rng.integers(0, 256, (37, 53), dtype=np.uint8),                                         # noise, odd size
             np.full((1, 1), 4, np.uint8), np.zeros((3, 500), np.uint8)]
    for a in cases:
        data = encode_png_gray8(a)
        assert np.array_equal(oio.png_decode_gray8(data), a)
        try:
            from PIL import Image
        except ImportError:
            continue
        im = Image.open(io.BytesIO(data))
        assert im.mode == 'L' and im.size == (a.shape[1], a.shape[0]) and np.array_equal(np.asarray(im), a)
    with pytest.raises(ValueError):
        encode_png_gray8(np.zeros((4, 4), np.int64))


def test_native_label_writer_host_tensors(tmp_path):
    """The native writer (C++ threads + zlib inside the library) with host tensors: no GPU involved; files decode to the maps,
    back-pressure with a single staging slot, error reporting for an unwritable directory."""
    from mspl_amd.io import LabelWriter
    rng = np.random.default_rng(5)
    maps = torch.from_numpy(np.repeat(np.repeat(rng.integers(0, 5, (4, 9, 13), dtype=np.uint8), 5, 1), 5, 2))
    with LabelWriter(str(tmp_path / 'pred'), workers=3, max_inflight=1) as w:
        for b in range(6):
            w.submit(['/data/color/f%d_%d.png' % (b, i) for i in range(4)], maps)
    assert len(w.label_paths) == 24 and w.label_paths[5] == '%s/f1_1.png' % (tmp_path / 'pred')
    for i, p in enumerate(w.label_paths):
        assert np.array_equal(oio.png_decode_gray8(open(p, 'rb').read()), maps[i % 4].numpy())
    w2 = LabelWriter(str(tmp_path / 'gone'), workers=1)
    os.rmdir(str(tmp_path / 'gone'))
    w2.submit(['a.png'], maps[:1])
    with pytest.raises(RuntimeError, match='file write'):
        w2.close()
    with pytest.raises(RuntimeError, match='closed'):
        w2.submit(['b.png'], maps[:1])


def test_image_list_round_trip(tmp_path):
    from mspl_amd.io import read_image_list, update_image_list
    files = []
    for n in ('a_color.png', 'a_label.png', 'a_depth.png', 'b_color.png', 'b_label.png', 'b_depth.png'):
        p = tmp_path / n
        p.write_bytes(b'x')
        files.append(str(p))
    lst = str(tmp_path / 'tgt_train.lst')
    update_image_list(lst, files[0::3], files[1::3])
    assert open(lst).read() == '%s,%s\n%s,%s\n' % (files[0], files[1], files[3], files[4])       # uest_seg_multi_os.py:726
    assert read_image_list(lst) == (files[0::3], files[1::3], [])
    update_image_list(lst, files[0::3], files[1::3], files[2::3])
    assert read_image_list(lst, use_depth=True) == (files[0::3], files[1::3], files[2::3])
    os.remove(files[4])
    with pytest.raises(AssertionError):
        read_image_list(lst)


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(IMAGEIO_CASES))
def test_device_transforms_bit_exact(name, golden):
    from mspl_amd.io import Preprocessor
    hs, ws, size, seed, norm, flip, with_depth = IMAGEIO_CASES[name]
    N = 3
    imgs = [synth_image_u8(hs, ws, seed + 100 * i) for i in range(N)]
    imgs[0] = synth_image_u8(hs, ws, seed)                                 # image 0 = the golden's image
    flips = [flip, not flip, flip]                                         # per-image flags
    rgb = torch.from_numpy(np.stack([im[0] for im in imgs]))
    lab = torch.from_numpy(np.stack([im[1] for im in imgs]))
    dep = torch.from_numpy(np.stack([im[2] for im in imgs])) if with_depth else None
    pre = Preprocessor(size=size, normalize=norm)
    x, y, d = pre(rgb, lab, dep, flip=torch.tensor(flips))
    assert x.shape == (N, 3, size[1], size[0]) and x.dtype == torch.float32 and y.dtype == torch.int64
    g = golden('imageio')
    assert np.array_equal(_sha(x[0].cpu().numpy()), g[name + '.rgb_sha'])
    assert np.array_equal(_sha(y[0].cpu().numpy()), g[name + '.label_sha'])
    if with_depth:
        assert np.array_equal(_sha(d[0].cpu().numpy()), g[name + '.depth_sha'])
    for i in range(N):                                                     # every image against the oracle
        t, lt, dt = oio.val_transform(imgs[i][0], imgs[i][1], imgs[i][2] if with_depth else None, size=size, normalise=norm,
                                      flip=flips[i])
        assert np.array_equal(x[i].cpu().numpy(), t) and np.array_equal(y[i].cpu().numpy(), lt)
        if with_depth:
            assert np.array_equal(d[i].cpu().numpy(), dt)
    with pytest.raises(RuntimeError, match='uint8'):
        pre(rgb.float())


@pytest.mark.gpu
def test_label_writer_async(tmp_path):
    from mspl_amd.io import LabelWriter, read_image_list, update_image_list
    rng = np.random.default_rng(11)
    maps = [torch.from_numpy(np.repeat(np.repeat(rng.integers(0, 5, (4, 32, 60), dtype=np.uint8), 8, 1), 8, 2)) for _ in range(3)]
    names = [['/data/greenhouse/color/img_%d_%d.jpg' % (b, i) for i in range(4)] for b in range(3)]
    with LabelWriter(str(tmp_path / 'pred'), workers=3) as w:
        for n, m in zip(names, maps):
            dev = m.cuda()
            w.submit(n, dev)
            dev.zero_()                                  # submit() snapshots: the caller may reuse its buffer at once
    images, labels = w.image_paths, w.label_paths
    assert images == sum(names, []) and labels[0] == '%s/img_0_0.png' % (tmp_path / 'pred')
    for b in range(3):
        for i in range(4):
            got = oio.png_decode_gray8(open(labels[b * 4 + i], 'rb').read())
            assert np.array_equal(got, maps[b][i].numpy())
    lst = str(tmp_path / 'tgt_train.lst')
    update_image_list(lst, images, labels)
    assert read_image_list(lst, check_files=False) == (images, labels, [])


@pytest.mark.gpu
@pytest.mark.parametrize('use_graph', [False, True])
def test_generate_pseudo_label_multi_model_end_to_end(tmp_path, use_graph):
    """The whole reference function (uest_seg_multi_os.py:832-956): uint8 frames -> device Resize+Normalize -> three source
    models -> LUT -> merge -> PNG files + tgt_train.lst + class weights; every stage checked against the oracle."""
    import argparse
    from mspl_amd import models, uest
    from mspl_amd.io import Preprocessor, read_image_list
    from oracle import labels as olab
    from oracle import net as onet
    from tests.synth import synth_state_dict
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    specs = [(13, 'camvid', 'camvid', 61), (20, 'city', 'cityscapes', 62), (5, 'greenhouse', 'forest', 63)]
    ms, sds = [], []
    for C_, ds, _, seed in specs:
        m = models.ESPDNetwithUncertaintyEstimation(a, classes=C_, dataset=ds, fix_pyr_plane_proj=True)
        sd = synth_state_dict(m.state_dict(), seed)
        m.load_state_dict(sd)
        ms.append(m)
        sds.append(sd)
    pre = Preprocessor(size=(64, 48))
    frames = [np.stack([synth_image_u8(72, 96, 70 + 4 * b + i)[0] for i in range(2)]) for b in range(5)]     # 5 batches: the 3 lanes wrap
    names = [['/d/color/f_%d_%d.jpg' % (b, i) for i in range(2)] for b in range(5)]
    loader = [(pre(torch.from_numpy(f))[0], None, n, 0.0) for f, n in zip(frames, names)]
    lst, cw = uest.generate_pseudo_label_multi_model(ms, [s[2] for s in specs], loader, str(tmp_path), use_graph=use_graph)
    images, labels, _ = read_image_list(lst, check_files=False)
    assert images == sum(names, []) and all(os.path.isfile(p) for p in labels)
    hist = np.zeros(5)
    for b in range(5):
        for i in range(2):
            x = torch.from_numpy(oio.val_transform(frames[b][i], size=(64, 48))[0])[None]
            srcs = []
            for sd, (_, _, os_data, _) in zip(sds, specs):
                with torch.no_grad():
                    main, aux = onet.espdnet_ue_forward(sd, x)
                prob, _ = olab.get_output(main, aux)
                srcs.append(olab.to_greenhouse(olab.argmax_labels(prob)[0], os_data))
            want = olab.merge_outputs(np.array(srcs), 5, 'all')
            got = oio.png_decode_gray8(open(labels[b * 2 + i], 'rb').read())
            # argmax ties between near-equal logits may flip a pixel between fp32 summation orders: allow a handful
            assert (got != want).mean() < 2e-3
            hist += np.bincount(got.ravel(), minlength=5)[:5]
    ref_w = olab.class_weights_from_histogram(hist)
    assert np.allclose(cw.cpu().numpy(), ref_w.astype(np.float32), rtol=1e-6)


@pytest.mark.gpu
def test_generate_pseudo_label_batches_per_launch(tmp_path):
    """Two consecutive loader batches per launch (PipelinedLabelPass group) with a ragged loader -- five batches of two and one
    of one image -- writes byte-identical label files, the same list and the same class weights as one batch per launch."""
    import argparse
    from mspl_amd import models, uest
    from mspl_amd.io import Preprocessor
    from tests.synth import synth_state_dict
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    ms = []
    for C_, ds, seed in [(13, 'camvid', 61), (5, 'greenhouse', 63)]:
        m = models.ESPDNetwithUncertaintyEstimation(a, classes=C_, dataset=ds, fix_pyr_plane_proj=True)
        m.load_state_dict(synth_state_dict(m.state_dict(), seed))
        ms.append(m)
    pre = Preprocessor(size=(64, 48))
    sizes = [2, 2, 2, 2, 2, 1]
    frames = [np.stack([synth_image_u8(72, 96, 90 + 4 * b + i)[0] for i in range(n)]) for b, n in enumerate(sizes)]
    names = [['/d/color/g_%d_%d.jpg' % (b, i) for i in range(n)] for b, n in enumerate(sizes)]
    outs = []
    for grp in (1, 2):
        d = tmp_path / ('g%d' % grp)
        d.mkdir()
        loader = [(pre(torch.from_numpy(f))[0], None, n, 0.0) for f, n in zip(frames, names)]
        lst, cw = uest.generate_pseudo_label_multi_model(ms, ['camvid', 'forest'], loader, str(d), use_graph=True, in_flight=2,
                                                         batches_per_launch=grp)
        files = sorted(os.listdir(str(d / 'pred')))
        outs.append((open(lst).read().replace(str(d), ''), cw.cpu().numpy(), {f: open(str(d / 'pred' / f), 'rb').read() for f in files}))
    assert outs[0][0] == outs[1][0] and len(outs[0][2]) == sum(sizes)
    assert np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2]
