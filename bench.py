#!/usr/bin/env python3
"""bench.py -- images/sec of the pseudo-label hot path on MI355X.

Workload (BASELINE.json configs[1]): ESPDNet-UE s=2.0, 13 classes, single-source pseudo-label generation
(forward -> pred+0.5*aux -> argmax + KL uncertainty map -> class histogram), batch 16 of synthetic
CamVid-shaped inputs.  "480x360" images are resized to the script's --crop-size before the network exactly
as the reference's loaders do: the network input is 16x3x288x480 fp32.
One step = one batch through the whole path; inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     dominant kernel = eesp_dw_exp (K2 + K3 of the stride-1 EESP blocks in one launch): matrix FLOPs per launch /
               mean launch time against the fp32 MFMA peak (+ its bytes against the HBM roof), measured with HIP events on the
               launch stream inside this process; roofline_k2 = the standalone K2 launches that remain (HBM).
  cpu_baseline the CPU oracle (oracle/, a port of the reference's torch path) timed on the host cores on a
               bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 16
H, W = 288, 480
CLASSES = 13
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
PATH_BYTES_PER_IMAGE = 356.9e6   # whole forward, ESPDNet-UE s=2.0 C=13 at 288x480


def k2_algorithmic_bytes(model, n_img, h, w):
    """Sum over the EESP blocks the forward executes of 4*n*(H*W + 4*Ho*Wo) bytes per image."""
    from mspl_amd.layers import EESP
    total, launches = 0, 0
    # walk the encoder exactly as _encode does: resolution after level1 is h/2
    b, d = model.base_net, getattr(model, 'depth_base_net', None)
    seq = [(b.level2_0.eesp, 2), (b.level3_0.eesp, 4), (b.level3[0], 8)]
    tail = d.level3 if d is not None else b.level3
    seq += [(tail[i], 8) for i in range(1, len(b.level3))]
    seq += [(b.level4_0.eesp, 8)] + [(m, 16) for m in b.level4]
    for m, div in seq:
        assert isinstance(m, EESP)
        n = m.proj_1x1.conv.out_channels
        hi, wi = h // div, w // div
        ho, wo = (hi - 1) // m.stride + 1, (wi - 1) // m.stride + 1
        total += 4 * n * (hi * wi + 4 * ho * wo)
        launches += 1
    return total * n_img, launches


def cpu_baseline(sd, shape, seconds_budget=20.0):
    """Oracle (CPU port) on the host cores: forward + softmax/KLD + argmax + histogram, bounded sample."""
    import numpy as np
    import torch
    from oracle import labels as olab, net as onet
    # the GPU box gives one GPU's share of the host: 16 cores (more threads than that only oversubscribe)
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    nb = 4
    x = torch.randn((nb,) + tuple(shape[1:]))
    sd = {k: v.cpu() for k, v in sd.items()}

    def one():
        with torch.no_grad():
            main, aux = onet.espdnet_ue_forward(sd, x)
            prob, kld = olab.get_output(main, aux)
            lab = olab.argmax_labels(prob)
            return olab.class_histogram(lab)
    one()  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += nb
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 192:          # ~10 s of wall time on 16 host cores
            break
    return {'value': round(n / el, 3), 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': '%d images (batches of %d, %dx%d), torch-CPU oracle, %d threads' % (n, nb, shape[2], shape[3], cores)}


def aspp_head_rate(dev, iters=10):
    """BASELINE configs[4]: the DeepLabv3 ASPP_Bottleneck head (three dense dilated 3x3 convolutions 2048 -> 256, K13) on
    16 x 2048 x 32 x 64 (1024x512 at output stride 16), hipGraph replay; the MFMA-bound piece of the path.  Extra field."""
    import torch
    from mspl_amd import aspp
    from tests.synth import synth_state_dict
    m = aspp.ASPP_Bottleneck(num_classes=20)
    m.load_state_dict(synth_state_dict(m.state_dict(), 0))
    m = m.to(dev).eval()
    x = torch.randn(BATCH, 2048, 32, 64, device=dev)
    with torch.no_grad():
        for _ in range(2):
            m(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            m(x)
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
    macs = BATCH * 32 * 64 * (2048 * 256 * 28 + 1280 * 256 + 256 * 20) + BATCH * 2048 * 256
    tf = 2 * macs / dt / 1e12
    return {'value': round(BATCH / dt, 1), 'unit': 'images/s', 'ms_per_batch': round(dt * 1e3, 3),
            'workload': 'ASPP_Bottleneck(num_classes=20), 16 x 2048 x 32 x 64 fp32 (1024x512 @ OS16), hipGraph replay',
            'roofline': {'bound': 'mfma', 'kernel': 'dense_conv_mfma_kernel (K13)', 'achieved': round(tf, 1), 'peak': 157.3,
                         'unit': 'TFLOP/s', 'frac': round(tf / 157.3, 4)}}


def loader_io_rate(dev, iters=20):
    """SURVEY 8f-1, the steps on either side of the path: (1) Resize(480x256)+Normalize of a batch of 16 decoded 360x480 uint8
    frames on the device (two launches, Pillow's fixed-point arithmetic, bit-exact) with Pillow + numpy on the host timed
    beside it (the reference's own loader path, one image at a time); (2) the asynchronous PNG writer draining 16 label
    maps (256x480 uint8).  Extra field."""
    import tempfile
    import numpy as np
    import torch
    from mspl_amd.io import LabelWriter, Preprocessor
    from tests.synth import synth_image_u8
    frames = np.stack([synth_image_u8(360, 480, 900 + i)[0] for i in range(BATCH)])
    src = torch.from_numpy(frames).to(dev)
    pre = Preprocessor(size=(480, 256))
    for _ in range(3):
        pre(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        pre(src)
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / 1e3 / iters
    alg = BATCH * (360 * 480 * 3 + 3 * 256 * 480 * 4)               # uint8 frame in, fp32 NCHW out
    out = {'device_transform': {'value': round(BATCH / dt, 1), 'unit': 'images/s', 'us_per_batch': round(dt * 1e6, 1),
                                'workload': '16 x 360x480x3 uint8 -> Resize((480,256), BILINEAR) -> Normalize -> 16x3x256x480 fp32',
                                'roofline': {'bound': 'hbm', 'achieved': round(alg / dt / 1e9, 1), 'peak': HBM_PEAK_GBS,
                                             'unit': 'GB/s', 'frac': round(alg / dt / 1e9 / HBM_PEAK_GBS, 4)}}}
    try:
        from PIL import Image
        mean = np.asarray([0.485, 0.456, 0.406], np.float32)[:, None, None]
        std = np.asarray([0.229, 0.224, 0.225], np.float32)[:, None, None]
        t0 = time.perf_counter()
        for f in frames:
            r = np.asarray(Image.fromarray(f).resize((480, 256), Image.BILINEAR))
            ((r.transpose(2, 0, 1).astype(np.float32) / np.float32(255)) - mean) / std
        out['host_pillow'] = {'value': round(BATCH / (time.perf_counter() - t0), 1), 'unit': 'images/s', 'cores': 1,
                              'kind': 'reference dependency (Pillow resize + numpy normalise, per image)'}
    except ImportError:
        pass
    labels = torch.from_numpy(np.stack([synth_image_u8(256, 480, 950 + i)[1] % 5 for i in range(BATCH)])).to(dev)
    names = ['/d/color/frame_%03d.jpg' % i for i in range(BATCH)]
    with tempfile.TemporaryDirectory() as d:
        w = LabelWriter(d, workers=8)
        t0 = time.perf_counter()
        for _ in range(2):                                # warm-up: pinned staging buffers are allocated once
            w.submit(names, labels)
        w._retire('all')
        t0 = time.perf_counter()
        for _ in range(8):
            w.submit(names, labels)
        t_submit = time.perf_counter() - t0
        w.close()
        t_all = time.perf_counter() - t0
    out['label_writer'] = {'value': round(8 * BATCH / t_all, 1), 'unit': 'images/s', 'workers': 8,
                           'submit_us_per_batch': round(t_submit / 8 * 1e6, 1),
                           'note': 'PNG encode + file write on native worker threads; submit() is what the label loop waits for'}
    # (3) the whole chain, host buffers in / files out: pinned uint8 frames -> H2D (PCIe) -> Resize+Normalize -> 13-class label pass
    #     (hipGraph) -> asynchronous D2H + PNG files.  This is the PCIe-inclusive rate; `value` above is the device-resident one.
    from mspl_amd import models, uest
    from tests.synth import synth_state_dict
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=13, dataset='camvid', fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(m.state_dict(), 0))
    lp = uest.PipelinedLabelPass(lambda: uest.SelfLabelPass(m, classes=13, device=dev, use_graph=True, with_kld=False), depth=3,
                                 device=dev, group=2)
    pinned = torch.from_numpy(frames).pin_memory()
    pre288 = Preprocessor(size=(480, 288))
    from mspl_amd.io import default_writer_workers
    workers = default_writer_workers()                         # PNG encoding is the host-side limit of the chain: 12 of the box's 16 cores
    # the reference function itself (uest_seg_multi_os.py:730-830) through its drop-in: loader of decoded uint8 frames -> transform into
    # the lane's input slot -> label pass -> PNG files -> tgt_train.lst -> class weights.  The timed region is the whole call.
    def loader(nb):
        for _ in range(nb):
            yield pinned, None, names, 0.0
    with tempfile.TemporaryDirectory() as d:
        uest.generate_pseudo_label(m, loader(12), d, classes=13, writer_workers=workers, _label_pass=lp, transform=pre288)   # warm-up: graph capture on every lane
        torch.cuda.synchronize()
        nb = 288                                              # ~0.35 s: long enough that pipeline fill / drain and the clock ramp stop mattering
        t0 = time.perf_counter()
        lst, cw = uest.generate_pseudo_label(m, loader(nb), d, classes=13, writer_workers=workers, _label_pass=lp, transform=pre288)
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        assert sum(1 for _ in open(lst)) == nb * BATCH and bool(torch.isfinite(cw).all())
        # the GPU side alone (no writer): the same loop body without the PNG files
        xs = lp.static_inputs((BATCH, 3, 288, 480))
        t0 = time.perf_counter()
        for _ in range(nb):
            with torch.cuda.stream(lp.next_stream):
                x = pre288(pinned, out=xs[lp.next_lane])[0]
            lp(x, on_lane=True)
        for _ in lp.flush(on_lane=True):
            pass
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t0
    out['end_to_end'] = {'value': round(nb * BATCH / t_all, 1), 'unit': 'images/s', 'gpu_side_images_per_s': round(nb * BATCH / t_gpu, 1),
                         'writer_workers': workers, 'function': 'mspl_amd.uest.generate_pseudo_label (uest_seg_multi_os.py:730-830), whole call timed',
                         'workload': 'pinned uint8 360x480 frames -> H2D -> Resize(480x288)+Normalize written into the lane\'s input slot -> ESPDNet-UE C=13 '
                                     'label pass (3 launches in flight, 2 batches per launch) -> async D2H + PNG files -> tgt_train.lst + class weights, %d batches of %d' % (nb, BATCH)}
    return out


def three_source_rate(dev, iters=24):
    """BASELINE configs[2], label half: three ESPDNet-UE source models (13 / 20 / 5 classes: CamVid, Cityscapes, Forest shapes)
    on the same 16 x 3 x 256 x 480 batch -> id LUTs -> merge_outputs('all') -> 5-bin histogram (uest_seg_multi_os.py:898-921),
    hipGraph replay, with one and with three batches in flight.  Extra field."""
    import torch
    from mspl_amd import models, uest
    from tests.synth import synth_state_dict
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    x = torch.randn(BATCH, 3, 256, 480, generator=torch.Generator().manual_seed(21)).to(dev)
    nets, datas = [], ['camvid', 'cityscapes', 'forest']
    for i, (C, ds) in enumerate([(13, 'camvid'), (20, 'city'), (5, 'forest')]):
        m = models.ESPDNetwithUncertaintyEstimation(a, classes=C, dataset=ds, fix_pyr_plane_proj=True)
        m.load_state_dict(synth_state_dict(m.state_dict(), i))
        nets.append(m)
    out = {'workload': 'BASELINE configs[2] label half: 3 ESPDNet-UE s=2.0 sources (C=13/20/5) -> LUT -> merge(all) -> histogram, '
                       '16 x 3 x 256 x 480 fp32, hipGraph replay', 'unit': 'images/s'}
    for depth in (1, 3):
        grp = 2 if depth > 1 else 1                     # two consecutive batches per launch, as in the headline
        plp = uest.PipelinedLabelPass(lambda: uest.PseudoLabelPass(nets, datas, merge_label_policy='all', device=dev, use_graph=True),
                                      depth=depth, device=dev, group=grp)
        for _ in range(2 * depth * grp + 2):
            plp(x)
        list(plp.flush())
        samples = []
        for _ in range(3):                              # median of three repetitions of `iters` batches, fill and drain included
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                plp(x)
            list(plp.flush())
            torch.cuda.synchronize()
            samples.append((time.perf_counter() - t0) / iters)
        dt = sorted(samples)[1]
        out['in_flight_%d' % depth] = {'value': round(BATCH / dt, 1), 'ms_per_batch': round(dt * 1e3, 3),
                                       'ms_per_batch_min_max': [round(min(samples) * 1e3, 3), round(max(samples) * 1e3, 3)]}
        hist = plp.hist.cpu().tolist()
        del plp
    out['value'] = out['in_flight_3']['value']
    out['histogram_pixels_per_batch'] = int(sum(hist)) // (3 * iters + 2 * 3 * 2 + 2)
    # 951 MB of algorithmic activation traffic per image for the three forwards + merge at 256x480
    out['path_roofline'] = {'algorithmic_bytes_per_image': 951e6, 'achieved': round(951e6 * out['value'] / 1e9, 1), 'peak': HBM_PEAK_GBS,
                            'unit': 'GB/s', 'frac': round(951e6 * out['value'] / 1e9 / HBM_PEAK_GBS, 4)}
    return out


def cityscapes_rate(dev, iters=12):
    """The label pass away from the tuned shape: ESPDNet-UE s=2.0 C=20 on Cityscapes-shaped 16 x 3 x 512 x 1024 batches
    (train_espdnetue_city.sh:6 trains at 512x256 and evaluates at 1024x512; BASELINE.md section 3: 1 392.2 MB of algorithmic
    activation traffic per image).  64 columns at level 4 and 128 at level 3: the stride-1 EESP blocks take the fused K2 + K3
    launch with one row per band / two half-row bands per row (csrc/eesp_exp.hip kinds 5 / 6).  hipGraph replay, three launches in
    flight, one batch per launch (a batch of 16 is already four times the pixels of the headline's).  Extra field."""
    import torch
    from mspl_amd import models, ops, uest
    from tests.synth import synth_state_dict
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=20, dataset='city', fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(m.state_dict(), 4))
    x = torch.randn(BATCH, 3, 512, 1024, generator=torch.Generator().manual_seed(22)).to(dev)
    bytes_img = 1392.2e6
    out = {'workload': 'ESPDNet-UE s=2.0 C=20 single-source label pass, 16 x 3 x 512 x 1024 fp32 (Cityscapes shape), hipGraph replay',
           'unit': 'images/s',
           'fused_eesp_launch': bool(ops.eesp_dw_exp_fits((BATCH, 128, 32, 64), [1, 1, 2, 3]) and
                                     ops.eesp_dw_exp_fits((BATCH, 64, 64, 128), [1, 2, 3, 4]))}
    for depth in (1, 3):
        plp = uest.PipelinedLabelPass(lambda: uest.SelfLabelPass(m, classes=20, device=dev, use_graph=True, with_kld=False),
                                      depth=depth, device=dev, group=1)
        for _ in range(2 * depth + 1):
            plp(x)
        list(plp.flush())
        samples = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                plp(x)
            list(plp.flush())
            torch.cuda.synchronize()
            samples.append((time.perf_counter() - t0) / iters)
        dt = sorted(samples)[1]
        out['in_flight_%d' % depth] = {'value': round(BATCH / dt, 1), 'ms_per_batch': round(dt * 1e3, 3),
                                       'path_roofline_frac': round(bytes_img * BATCH / dt / 1e9 / HBM_PEAK_GBS, 4)}
        pixels = int(plp.hist.sum())
        del plp
    out['value'] = out['in_flight_3']['value']
    out['pixels_counted_ok'] = pixels == (3 * iters + 2 * 3 + 1) * BATCH * 512 * 1024
    out['path_roofline'] = {'algorithmic_bytes_per_image': bytes_img, 'achieved': round(bytes_img * out['value'] / 1e9, 1),
                            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(bytes_img * out['value'] / 1e9 / HBM_PEAK_GBS, 4),
                            'source': 'BASELINE.md section 3 (SURVEY 8(d) accounting at 512x1024, C=20)'}
    return out


SUPERVISED_BYTES_PER_IMAGE = 947.7e6     # tools/train_bytes.py 13 288 480: the same accounting for C=13 at 288x480
SUPERVISED_BN_STAT_BYTES_PER_IMAGE = 316.7e6     # ... and what batch statistics need on top: per BatchNorm one read of its input (forward) and one
                                                 # pass over (gradient, input) for the two channel sums (backward)
EVAL_BYTES_PER_IMAGE = 306.7e6           # forward of the C=5 model at 256x480 (305.7 MB, tools/train_bytes.py) + the int64 labels (8 B / pixel)
TRAIN_BYTES_PER_IMAGE = 820.5e6          # DESIGN.md section 7 / tools/train_bytes.py: forward (305.7 MB, = SURVEY 8(d)'s 306.9) + data-gradient + weight-gradient
                                         # passes per image in SURVEY 8(d)'s accounting (weighted layers 3x their forward bytes, weightless ones 2x), C=5, 256x480


def train_step_build(dev, rank=0):
    """Everything of the uest train step that can fail on one rank alone (model, first eager step, graph capture), WITHOUT a
    collective: under N > 1 it runs inside mspl_amd.dist.local_only()."""
    import torch
    from mspl_amd import models, training
    from tests.synth import synth_state_dict
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(m.state_dict(), 9))
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(7 + rank)           # every rank trains on its own shard (data parallel)
    x = torch.randn((BATCH, 3, 256, 480), generator=g).to(dev)
    y = torch.randint(0, 5, (BATCH, 256, 480), generator=g).to(dev)
    # the batch of 16 runs as 2 concurrent micro-batch graphs (same step: frozen BN, mean loss, atomic gradient sinks).
    # Round 5: 1 / 2 / 4 lanes = 5.91 / 5.72-5.80 / 5.79-5.80 ms on the final kernels -- two need two hardware queues, not four
    lanes = int(os.environ.get('MSPL_TRAIN_LANES', '2'))
    step = training.GraphedTrainStep(m, x, y, torch.ones(5), ignore_idx=4, lanes=lanes)
    return step, x, y


def train_step_time(step, x, y, dev, iters=10, world=1, repeats=3):
    """Timed part: `iters` steps between fences, `repeats` times, the median repetition (N > 1: one flat-bucket gradient all-reduce
    per step inside the loop, MAX over ranks per repetition)."""
    import torch
    import torch.distributed as dist

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(2):
        step(x, y)
    samples = []
    for _ in range(repeats):
        fence()
        t0 = time.perf_counter()
        for _ in range(iters):
            loss = step(x, y)
        fence()
        samples.append((time.perf_counter() - t0) / iters)
    in_sync = None
    if world > 1:
        t = torch.tensor(samples, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        samples = [float(v) for v in t.tolist()]
        # data-parallel invariant: identical initial weights + averaged gradients => identical weights on every rank
        chk = step.optimizer.flat_p.double().sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool((hi - lo).abs().item() == 0.0)
    dt = sorted(samples)[len(samples) // 2]
    achieved = TRAIN_BYTES_PER_IMAGE * BATCH / dt / 1e9
    return {'value': round(BATCH * world / dt, 1), 'unit': 'images/s', 'ms_per_step': round(dt * 1e3, 3), 'steps': iters, 'repeats': repeats,
            'ms_per_step_min_max': [round(min(samples) * 1e3, 3), round(max(samples) * 1e3, 3)],
            'n_gpus': world, 'global_batch': BATCH * world,
            'micro_batch_lanes': step.lanes, 'lane_overlap': getattr(step, 'lane_overlap', None),
            'workload': 'uest train step, ESPDNet-UE s=2.0 C=5, bs=16/GPU x 3 x 256 x 480 fp32, hipGraph replay (%d concurrent micro-batch graphs) + ' % step.lanes +
                        ('one flat-bucket gradient all-reduce (%d floats, RCCL) + ' % step.optimizer.flat_g.numel() if world > 1 else '') +
                        'Adam kernel',
            # forward + backward against the HBM roof: every weighted layer's input and output are moved once by the forward, once by the
            # data gradient, once by the weight gradient (weightless layers twice; BN / PReLU / add / cat fused = 0): tools/train_bytes.py
            'roofline': {'bound': 'hbm', 'algorithmic_bytes_per_image': TRAIN_BYTES_PER_IMAGE, 'achieved': round(achieved, 1),
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4)},
            'weights_identical_across_ranks': in_sync, 'loss_finite': bool(torch.isfinite(loss))}


def train_step_rate(dev, iters=10, world=1, rank=0):
    """BASELINE configs[2]'s other half, reported beside the headline: the uest train step (frozen-BN forward, fused
    KLD + uncertainty-weighted CE, backward, Adam) of the 5-class target model, bs=16 at 256x480, as hipGraph replays
    (+ the Adam kernel) per step.  Extra field; `value` stays the label-pass metric."""
    step, x, y = train_step_build(dev, rank)
    return train_step_time(step, x, y, dev, iters, world)


def eval_step_rate(dev, iters=20):
    """The evaluation step of the loop (utilities/train_eval_seg.py:249-324 val_seg_ue; uest_seg_multi_os.py:1150-1200 test()):
    forward -> out + 0.5*aux -> weighted cross entropy -> MIOU areas of the 5-class target model, bs=16 at 256x480, one hipGraph
    replay per batch (forward_lowres + the fused eval epilogue; full-resolution logits are never written).  Extra field."""
    import torch
    from mspl_amd import evaluation, models
    from tests.synth import synth_state_dict
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(m.state_dict(), 9))
    g = torch.Generator().manual_seed(17)
    x = torch.randn((BATCH, 3, 256, 480), generator=g).to(dev)
    y = torch.randint(0, 5, (BATCH, 256, 480), generator=g).to(dev)
    res = {}
    for name, ep, n in (('one_in_flight', evaluation.EvalPass(m, 5, class_weights=torch.ones(5), ignore_idx=4, aux_weight=0.5, device=dev,
                                                             use_graph=True), iters),
                        ('lanes_3', evaluation.PipelinedEvalPass(m, 5, depth=3, class_weights=torch.ones(5), ignore_idx=4, aux_weight=0.5,
                                                                 device=dev), 3 * iters),
                        ('lanes_3x2', evaluation.PipelinedEvalPass(m, 5, depth=3, group=2, class_weights=torch.ones(5), ignore_idx=4,
                                                                   aux_weight=0.5, device=dev), 6 * iters)):
        for _ in range(12):
            ep(x, y)
        ep.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            ep(x, y)
        iou, loss = ep.result(reduce=False)                     # joins the lanes; one device-to-host copy of the sums
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res[name] = {'value': round(BATCH / dt, 1), 'ms_per_batch': round(dt * 1e3, 3), 'batches': n, 'loss_finite': bool(np_isfinite(loss)),
                     'pixels_counted': int(ep.sums()[2 * ep.K:3 * ep.K].sum().item())}
    best = res['lanes_3x2']
    for r_ in res.values():
        gbs = EVAL_BYTES_PER_IMAGE * r_['value'] / 1e9
        r_['roofline'] = {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4)}
    return {'value': best['value'], 'unit': 'images/s', 'ms_per_batch': best['ms_per_batch'], 'batches': best['batches'],
            'roofline': dict(best['roofline'], algorithmic_bytes_per_image=EVAL_BYTES_PER_IMAGE,
                             accounting='SURVEY 8(d): the forward\'s convolutions as in + out (305.7 MB, tools/train_bytes.py 5 256 480) + the labels'),
            'workload': 'val_seg_ue step, ESPDNet-UE s=2.0 C=5, bs=16 x 3 x 256 x 480 fp32: forward + out+0.5*aux + weighted CE + MIOU '
                        'areas, hipGraph replays, 3 launches in flight x 2 consecutive batches per launch (what val_seg_ue runs; per-batch '
                        'loss means and meter updates as in the reference loop)',
            'loss_finite': best['loss_finite'], 'pixels_counted': best['pixels_counted'], 'one_in_flight': res['one_in_flight'],
            'one_batch_per_launch': res['lanes_3']}


def np_isfinite(v):
    import math
    return math.isfinite(float(v))


def supervised_step_rate(dev, iters=6):
    """one train_seg_ue iteration of the supervised source-model loop (model.train(): batch-statistics BatchNorm,
    CrossEntropy on main + 0.5*aux, flooding, SGD with two learning-rate groups), ESPDNet-UE C=13, bs=16 at 288x480 (the
    CamVid crop of train_espdnetue_camvid.sh), zero_grad + forward + loss + backward as one hipGraph replay.  Extra field."""
    import torch
    from mspl_amd import losses, models, supervised
    from tests.synth import synth_state_dict
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=13, dataset='camvid', fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(m.state_dict(), 10))
    m = m.to(dev).train()
    g = torch.Generator().manual_seed(8)
    x = torch.randn((BATCH, 3, 288, 480), generator=g).to(dev)
    y = torch.randint(0, 13, (BATCH, 288, 480), generator=g).to(dev)
    crit = losses.SegmentationLoss(n_classes=13, device=dev, ignore_idx=255)
    step = supervised.GraphedSupervisedStep(m, x, y, crit)
    for _ in range(2):
        loss, _ = step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        loss, _ = step(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    total_bytes = SUPERVISED_BYTES_PER_IMAGE + SUPERVISED_BN_STAT_BYTES_PER_IMAGE
    achieved = total_bytes * BATCH / dt / 1e9
    return {'value': round(BATCH / dt, 1), 'unit': 'images/s', 'ms_per_step': round(dt * 1e3, 3), 'steps': iters,
            'roofline': {'bound': 'hbm', 'algorithmic_bytes_per_image': total_bytes, 'achieved': round(achieved, 1),
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                         'frac_without_statistics_passes': round(SUPERVISED_BYTES_PER_IMAGE * BATCH / dt / 1e9 / HBM_PEAK_GBS, 4),
                         'accounting': 'tools/train_bytes.py 13 288 480: forward + data gradient + weight gradient of every weighted layer '
                                       '(947.7 MB / image, the train step\'s accounting, BatchNorm / PReLU = 0) + the batch-statistics '
                                       'passes this mode needs (316.7 MB / image: per BatchNorm one read of its input for the statistics and '
                                       'one pass over (gradient, input) for the backward\'s channel sums); rounds 3-4 quoted the first term alone '
                                       '(frac_without_statistics_passes)'},
            'workload': 'train_seg_ue iteration, ESPDNet-UE s=2.0 C=13 in train() (batch-statistics BN), bs=16 x 3 x 288 x 480 fp32, '
                        'CrossEntropy + flooding + SGD(2 lr groups), hipGraph replay + SGD kernels',
            'loss_finite': bool(torch.isfinite(loss))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-train', action='store_true', help='skip the extra train-step field')
    ap.add_argument('--no-three-source', action='store_true', help='skip the extra 3-source label pass field (BASELINE configs[2])')
    ap.add_argument('--no-io', action='store_true', help='skip the extra loader/writer field (SURVEY 8f-1)')
    ap.add_argument('--no-aspp', action='store_true', help='skip the extra ASPP-head field (BASELINE configs[4])')
    ap.add_argument('--no-bs64', action='store_true', help='skip the extra batch-64 K2 field (use for rocprofv3 --stats runs: '
                    'its launches would mix into the per-kernel averages)')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--in-flight', type=int, default=3, help='label passes (independent batches) in flight on the GPU; 1 = one '
                    'hipGraph replayed back to back on one stream (use it for rocprofv3 --stats runs: overlapping launches stretch '
                    'each other and the per-kernel averages stop describing the kernels)')
    ap.add_argument('--group', type=int, default=2, help='consecutive batches of 16 that one launch of a lane labels (PipelinedLabelPass '
                    'group): 2 = 32 images per launch, 3 lanes; images are independent, results are per batch')
    ap.add_argument('--repeats', type=int, default=5, help='the timed loop of --steps steps is run this many times inside one invocation '
                    '(each bracketed by barrier + synchronize); ms_per_step / value are the MEDIAN repetition, min / max are reported beside it')
    ap.add_argument('--profile-pass', action='store_true', help='only the timed label passes: no K2 re-issues, no extra fields '
                    '(for rocprofv3 --kernel-trace --stats: the CSV then holds in-pass launches only; tools/per_kernel.py)')
    ap.add_argument('--profile-batch', type=int, default=0, help='with --profile-pass only: images per launch (kernel scaling study; '
                    'the metric itself is always batch 16)')
    args = ap.parse_args()
    global BATCH
    if args.profile_pass and args.profile_batch > 0:
        BATCH = args.profile_batch

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one fresh process per GPU ourselves (torch.distributed.run, the
        # driver's own launch line) BEFORE this process touches the GPU, hand its output through and exit with its code.
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # "nccl" is RCCL on ROCm.  MSPL_BENCH_BACKEND=gloo rehearses the N>1 code path on a box with fewer GPUs than ranks (the
    # ranks then share devices round-robin; the numbers mean nothing, the control flow is the same).
    backend = os.environ.get('MSPL_BENCH_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1) if (world > 1 and backend != 'nccl') else (local_rank if world > 1 else 0)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        torch.cuda.set_device(dev_index)
        dist.init_process_group(backend, rank=rank, world_size=world)   # barrier, max-over-ranks timing, gradient all-reduce
    dev = torch.device('cuda', dev_index)
    torch.cuda.set_device(dev)

    import mspl_amd
    from mspl_amd import models, ops, uest
    from tests.synth import synth_state_dict

    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    model = models.ESPDNetwithUncertaintyEstimation(a, classes=CLASSES, dataset='camvid', fix_pyr_plane_proj=True)
    sd = synth_state_dict(model.state_dict(), 0)          # random-init weights of the named architecture
    model.load_state_dict(sd)
    shape = (BATCH, 3, H, W)
    g = torch.Generator().manual_seed(1234 + rank)        # every rank labels its own shard of the image list
    x = torch.randn(shape, generator=g).to(dev)

    depth = max(1, args.in_flight)
    group = max(1, args.group) if depth > 1 else 1
    plp = uest.PipelinedLabelPass(lambda: uest.SelfLabelPass(model, classes=CLASSES, device=dev, use_graph=not args.no_graph),
                                  depth=depth, device=dev, group=group)
    for _ in range(depth * group):
        plp(x)                                             # builds caches / captures one graph per lane
    list(plp.flush())
    if group > 1 and args.steps % group:                   # an odd tail is labelled by a one-batch launch: capture it now, not in the timed region
        for lane in plp.lanes:
            lane(x)
        torch.cuda.synchronize()
    xs = plp.static_inputs(shape)                          # every lane labels its own resident copy of the batch: no input copy
    for i, xi in enumerate(xs):
        if xi is not None:
            xi.copy_(x)
        else:
            xs[i] = x

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(steps):
        for _ in range(steps):
            plp(xs[plp.next_lane])
        list(plp.flush())

    # warm-up: at least W steps, rounded up to whole launches (a partly filled lane would be labelled by a shorter launch)
    run(-(-args.warmup // group) * group)
    repeats = max(1, 1 if args.profile_pass else args.repeats)
    samples = []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        run(args.steps)                                    # EXACTLY K steps, pipeline fill and drain included
        barrier()
        samples.append(time.perf_counter() - t0)
    if world > 1:
        t = torch.tensor(samples, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)           # every repetition: the slowest rank's time
        samples = [float(v) for v in t.tolist()]
    elapsed = sorted(samples)[len(samples) // 2]           # median repetition

    if args.profile_pass:
        if rank == 0:
            print(json.dumps({'metric': 'profile-pass (no roofline / extras)', 'value': round(BATCH * world * args.steps / elapsed, 2),
                              'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                              'batches_in_flight': depth, 'batches_per_launch': group}))
        if world > 1:
            dist.destroy_process_group()
        return

    # N > 1 (configs[3]: the same 3-source pipeline sharded over the GPUs): every rank runs the 3-source label pass and the
    # evaluation step on its own shard (no collective inside: only the slowest rank's time is reduced afterwards) and the
    # data-parallel train step (the gradient all-reduce over xGMI is the path's only data collective).  Whatever can fail on one
    # rank alone runs WITHOUT collectives first; the ranks then agree on success with one all-reduce, so a rank that raised never
    # leaves the others blocked in a collective it will not join.
    multi = {}
    if world > 1 and not args.no_train:
        from mspl_amd import dist as mdist

        def all_ok(ok):
            t = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        def per_rank(name, fn):
            res, err = None, None
            try:
                with mdist.local_only():
                    res = fn(dev)
            except Exception as e:      # noqa: BLE001
                err = repr(e)[:300]
            if not all_ok(err is None):
                multi[name] = {'error': err or 'another rank failed', 'skipped_on_all_ranks': True}
                return
            t = torch.tensor([BATCH / res['value']], device=dev, dtype=torch.float64)      # seconds per batch on this rank
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            res = dict(res, value=round(BATCH * world / float(t.item()), 1), n_gpus=world,
                       note='every rank on its own shard, no data-path collective; value = images of all ranks / slowest rank\'s time')
            multi[name] = res
        if not args.no_three_source:
            per_rank('three_source', three_source_rate)
        per_rank('eval_step', eval_step_rate)
        built, err = None, None
        try:
            with mdist.local_only():
                built = train_step_build(dev, rank)
        except Exception as e:      # noqa: BLE001
            err = repr(e)[:300]
        if all_ok(err is None):
            step_, x_, y_ = built
            # the set-up steps ran without the gradient exchange: put every rank back on rank 0's weights and optimizer state
            for t_ in (step_.optimizer.flat_p, step_.optimizer.m, step_.optimizer.v):
                dist.broadcast(t_, src=0)
            from mspl_amd import layers as _L
            _L.bump_param_epoch()
            try:
                multi['train_step'] = train_step_time(step_, x_, y_, dev, world=world)
            except Exception as e:      # noqa: BLE001
                multi['train_step'] = {'error': repr(e)[:300]}
        else:
            multi['train_step'] = {'error': err or 'another rank failed', 'skipped_on_all_ranks': True}

    # the same K steps with ONE pass in flight (lane 0 alone, back to back): the per-batch latency, reported beside the value
    single = None
    if depth > 1 and rank == 0:
        solo = uest.SelfLabelPass(model, classes=CLASSES, device=dev, use_graph=not args.no_graph)   # with its intra-pass branches
        solo(x)
        xs0 = solo.static_input(shape)
        if xs0 is not None:
            xs0.copy_(x)
        else:
            xs0 = x
        for _ in range(min(5, args.warmup)):
            solo(xs0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            solo(xs0)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        single = {'value': round(BATCH * args.steps / e1, 2), 'unit': 'images/s', 'ms_per_step': round(e1 / args.steps * 1e3, 4),
                  'note': 'one hipGraph (with its intra-pass parallel branches) replayed back to back on one stream = latency of a batch'}
        del solo
    x = xs[0]

    # ---- roofline of the dominant kernels: the EESP depthwise family.  Round 4: the ten stride-1 blocks run K2 + K3 as ONE launch
    # (ops.eesp_dw_exp, csrc/eesp_exp.hip: the dominant kernel of a pass, ~24 % of its kernel time), the three strided blocks keep
    # the standalone K2 launch (ops.eesp_dw_hff).  Both kinds are recorded in one eager pass (same tensors, same shapes), re-issued
    # REPS times back to back between one event pair (stream parked behind a spin kernel) for the isolated figure, and timed IN the
    # pass with an event pair around every launch.
    MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_32x32x2_f32; = the fp32 vector peak)
    calls = []              # (kind, fn, args, kwargs)
    real_k2, real_exp = ops.eesp_dw_hff, ops.eesp_dw_exp

    def record_k2(*a_, **kw):
        r = real_k2(*a_, **kw)
        calls.append(('k2', real_k2, a_, dict(kw, out=(r, 0))))
        return r

    def record_exp(*a_, **kw):
        r = real_exp(*a_, **kw)
        calls.append(('exp', real_exp, a_, kw))
        return r
    eager = uest.SelfLabelPass(model, classes=CLASSES, device=dev, use_graph=False)
    from mspl_amd import layers as L
    L.ops.eesp_dw_hff, L.ops.eesp_dw_exp = record_k2, record_exp
    try:
        with ops.launch_flags(throughput=True):        # the lanes' launch shapes
            eager(x)
        torch.cuda.synchronize()
    finally:
        L.ops.eesp_dw_hff, L.ops.eesp_dw_exp = real_k2, real_exp

    def call_bytes(c, mult=1):
        """Algorithmic bytes of one launch.  K2: 4*n*(H*W + 4*Ho*Wo) per image.  Fused K2 + K3: what the launch has to
        move -- read the reduced tensor (n) and the residual (4n), write the result (4n): 4*9n*H*W per image (the two launches it
        replaces: 4*17n*H*W with the residual, 4*13n*H*W by SURVEY 8d's convention of counting a convolution as in + out)."""
        kind, _, a_, kw = c
        n_, ch_, hi_, wi_ = a_[0].shape
        if kind == 'exp':
            # (+ n*H*W written when the launch also computes the next block's proj_1x1)
            return mult * 4 * n_ * (9 + (1 if kw.get('next_proj') is not None else 0)) * ch_ * hi_ * wi_
        st_ = a_[3] if len(a_) > 3 else kw.get('stride', 1)
        ho_, wo_ = (hi_ - 1) // st_ + 1, (wi_ - 1) // st_ + 1
        return mult * 4 * n_ * ch_ * (hi_ * wi_ + 4 * ho_ * wo_)

    def call_flops(c, mult=1):
        kind, _, a_, _ = c
        if kind != 'exp':
            return 0
        n_, ch_, hi_, wi_ = a_[0].shape
        # the grouped expansion: 4n outputs x n inputs per pixel; + the next block's grouped projection (n outputs x n inputs) when fused in
        return mult * 2 * n_ * hi_ * wi_ * (4 + (1 if c[3].get('next_proj') is not None else 0)) * ch_ * ch_

    REPS = 20
    iso_ms = []
    for _, fn, a_, kw in calls:
        for _ in range(2):
            fn(*a_, **kw)
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(2_000_000)
            e0.record()
            for _ in range(REPS):
                fn(*a_, **kw)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / REPS
            best = t if best is None or t < best else best
        iso_ms.append(best)
    k2_launches = len(calls)
    n_exp = sum(1 for c in calls if c[0] == 'exp')
    model_bytes, model_launches = k2_algorithmic_bytes(model, BATCH, H, W)
    family_note = ('%d of the %d launches are the fused K2 + K3 launches of the stride-1 blocks (levels 3 / 4), bounded by the SIMD issue '
                   'the fp32 MFMAs and the depthwise vector work share (measured: the two do not overlap on a SIMD), not by HBM; the other %d '
                   'are the strided blocks\' standalone K2 launches (streaming / direct form), HBM-side' % (n_exp, k2_launches, k2_launches - n_exp))
    if model_launches != k2_launches:
        family_note += '; recorded %d launches, the model walk gives %d' % (k2_launches, model_launches)

    # ---- the same launches timed IN the pass: one eager label pass, the stream parked behind a spin kernel (so the host's launch
    # cadence is out of the picture), a HIP event pair around every launch.  An event pair adds its own time to what it brackets;
    # that overhead is calibrated PER SHAPE on the isolated kernel: (event pair around ONE warm launch) - (back-to-back time per
    # launch of the same kernel, measured above), and subtracted.  This is the number rocprofv3's in-pass average must agree with:
    # inputs come from the producer kernel through L2 / Infinity Cache / HBM as in the real pass, not from 20 warm re-issues.
    def pair_overhead():
        ovh = []
        for (_, fn, a_, kw), b2b in zip(calls, iso_ms):
            ts = []
            for _ in range(5):
                fn(*a_, **kw)
                torch.cuda._sleep(2_000_000)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(*a_, **kw)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ovh.append(max(0.0, sorted(ts)[len(ts) // 2] - b2b))
        return ovh

    def family_in_pass(mult, ovh, passes=7):
        xin = x if mult == 1 else torch.cat([x] * mult, 0)
        per_pass = []
        rec = []

        def timed(fn):
            def run(*a_, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = fn(*a_, **kw)
                e1.record()
                rec.append((e0, e1))
                return r
            return run
        L.ops.eesp_dw_hff, L.ops.eesp_dw_exp = timed(real_k2), timed(real_exp)
        try:
            with L.side_streams(False), ops.launch_flags(throughput=True):
                eager(xin)                                  # allocator / caches warm for this batch size
                torch.cuda.synchronize()
                for _ in range(passes):
                    del rec[:]
                    torch.cuda._sleep(40_000_000)
                    eager(xin)
                    torch.cuda.synchronize()
                    per_pass.append([e0.elapsed_time(e1) for e0, e1 in rec])
        finally:
            L.ops.eesp_dw_hff, L.ops.eesp_dw_exp = real_k2, real_exp
        n = len(per_pass[0])
        return [sorted(pp[i] for pp in per_pass)[len(per_pass) // 2] - ovh[i] for i in range(n)]

    def summarise(ms, mult=1):
        """Per kind: launches, average launch time, bytes / flops over the summed time."""
        res = {}
        for kind in ('exp', 'k2', 'all'):
            idx = [i for i, c in enumerate(calls) if kind == 'all' or c[0] == kind]
            if not idx:
                continue
            t_s = sum(ms[i] for i in idx) * 1e-3
            by = sum(call_bytes(calls[i], mult) for i in idx)
            fl = sum(call_flops(calls[i], mult) for i in idx)
            res[kind] = {'launches': len(idx), 'avg_launch_us': round(t_s / len(idx) * 1e6, 3), 'gbs': round(by / t_s / 1e9, 1),
                         'tflops': round(fl / t_s / 1e12, 2), 'bytes_per_launch': int(by / len(idx)),
                         'per_launch_us': [round(ms[i] * 1e3, 2) for i in idx]}
        return res
    iso = summarise(iso_ms)
    in_pass = {}
    ovh = None
    if rank == 0:
        try:
            ovh = pair_overhead()
        except Exception as e_:      # noqa: BLE001
            in_pass[1] = {'error': repr(e_)[:200]}
        for mult in ((1, group) if group > 1 else (1,)) if ovh is not None else ():
            try:
                in_pass[mult] = summarise(family_in_pass(mult, ovh), mult)
                in_pass[mult]['event_pair_overhead_us'] = round(sum(ovh) / len(ovh) * 1e3, 3)
            except Exception as e_:      # noqa: BLE001
                in_pass[mult] = {'error': repr(e_)[:200]}

    # The standalone K2 launches at 4x the batch ("report K2 at bs=16 and bs=64").
    k2_ms64 = []
    if rank == 0 and not args.no_bs64:
        for kind, fn, a_, kw in calls:
            if kind != 'k2':
                continue
            xin64 = torch.cat([a_[0]] * 4, 0)
            kw64 = {k: v for k, v in kw.items() if k != 'out'}
            for _ in range(2):
                fn(xin64, *a_[1:], **kw64)
            torch.cuda.synchronize()
            best = None
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(2_000_000)
                e0.record()
                for _ in range(REPS):
                    fn(xin64, *a_[1:], **kw64)
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / REPS
                best = t if best is None or t < best else best
            k2_ms64.append(best)
            del xin64
    # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE passes of this same command, corrected as the MI355X
    # guide prescribes; tools/k2_traffic.py writes the summary).  bench.py cannot run rocprofv3 on itself.
    def newest(names):
        for n in names:
            if os.path.exists(os.path.join(ROOT, 'profiles', n)):
                return n
        return None
    fam_traffic = {}
    tname = newest(['r05_k2_hbm_traffic.json', 'r04_k2_hbm_traffic.json'])
    if tname:
        tj = json.load(open(os.path.join(ROOT, 'profiles', tname)))
        fam_traffic = {'exp': tj.get('fused_avg_traffic_bytes_per_launch'), 'k2': tj.get('standalone_avg_traffic_bytes_per_launch'),
                       'source': 'profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)' % tname}
    # measured HBM bytes of one whole pass (every kernel; tools/pass_traffic.py, same PMC recipe)
    path_traffic, path_traffic_src = None, None
    pname = newest(['r05_pass_hbm_traffic.json', 'r04_pass_hbm_traffic.json', 'r03_pass_hbm_traffic.json', 'r02_pass_hbm_traffic.json'])
    if pname:
        path_traffic = int(json.load(open(os.path.join(ROOT, 'profiles', pname)))['total_MB_per_image'] * 1e6)
        path_traffic_src = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE over one pass)' % pname
    # the same kernels' average duration in the committed rocprofv3 --kernel-trace --stats summary of `bench.py --profile-pass
    # --in-flight 1` (label passes only): the figure the live in-pass measurement has to agree with
    rocprof_us, rocprof_src = {}, None
    cname = newest(['r05_kernel_stats_inflight1.csv', 'r04_kernel_stats_inflight1.csv'])
    if cname:
        import csv
        acc_ = {'exp': [0.0, 0], 'k2': [0.0, 0]}
        for row in csv.DictReader(open(os.path.join(ROOT, 'profiles', cname))):
            kind = 'exp' if 'eesp_dw_exp_kernel' in row['Name'] else ('k2' if any(k in row['Name'] for k in ('eesp_dw_hff_kernel', 'eesp_dw_direct_kernel', 'eesp_dw_stream2_kernel')) else None)
            if kind:
                acc_[kind][0] += float(row['TotalDurationNs'])
                acc_[kind][1] += int(row['Calls'])
        rocprof_us = {k: round(v[0] / v[1] / 1e3, 3) for k, v in acc_.items() if v[1]}
        rocprof_src = 'profiles/%s' % cname

    def roof_exp(sm):
        return None if not sm or 'exp' not in sm else {
            'bound': 'mfma', 'achieved': sm['exp']['tflops'], 'peak': MFMA_PEAK_TF, 'unit': 'TFLOP/s',
            'frac': round(sm['exp']['tflops'] / MFMA_PEAK_TF, 4), 'avg_launch_us': sm['exp']['avg_launch_us'],
            'launches': sm['exp']['launches'], 'algorithmic_bytes_per_launch': sm['exp']['bytes_per_launch'],
            'hbm_view': {'achieved': sm['exp']['gbs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(sm['exp']['gbs'] / HBM_PEAK_GBS, 4)},
            'per_launch_us': sm['exp']['per_launch_us']}

    def roof_k2(sm):
        return None if not sm or 'k2' not in sm else {
            'bound': 'hbm', 'achieved': sm['k2']['gbs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(sm['k2']['gbs'] / HBM_PEAK_GBS, 4),
            'avg_launch_us': sm['k2']['avg_launch_us'], 'launches': sm['k2']['launches'],
            'algorithmic_bytes_per_launch': sm['k2']['bytes_per_launch'], 'per_launch_us': sm['k2']['per_launch_us']}

    if rank == 0:
        ip1 = in_pass.get(1, {})
        out = {
            'metric': 'images/sec pseudo-label gen, ESPDNet-UE s=2.0 480x360(->288x480) bs=16',
            'value': round(BATCH * world * args.steps / elapsed, 2),
            'unit': 'images/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'repeats': repeats,
            'ms_per_step_min_max': [round(min(samples) / args.steps * 1e3, 4), round(max(samples) / args.steps * 1e3, 4)],
            'timing_note': 'each repetition = %d steps between barrier + synchronize, pipeline fill and drain included; value / ms_per_step = '
                           'the median repetition' % args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: ESPDNet-UE s=2.0 C=13 single-source pseudo-label gen '
                                   '(forward + pred+0.5aux argmax + KL uncertainty + histogram), bs=16/GPU, '
                                   '16x3x288x480 fp32, hipGraph replay' + (' off' if args.no_graph else '') +
                                   ', %d launches in flight per GPU, %d consecutive batches of 16 per launch' % (depth, group),
                       'per_gpu_batch': BATCH, 'batches_in_flight': depth, 'batches_per_launch': group, 'input': [BATCH, 3, H, W],
                       'classes': CLASSES,
                       'sharding': 'image list sharded by rank, no data-path collective'},
            # `roofline` = the dominant kernel (the fused K2 + K3 launch) at its IN-PASS launch time; `roofline_k2` = the standalone K2
            # launches that remain (the strided blocks); `roofline_family` = all thirteen against their bytes, for continuity with
            # rounds 1-3 (whose `roofline` was the thirteen standalone K2 launches: 0.30 of the HBM roof)
            'roofline': dict(roof_exp(ip1 if 'exp' in ip1 else iso) or {}, **{
                'kernel': 'eesp_dw_exp (K2 + K3 of a stride-1 EESP block, and the NEXT block\'s proj_1x1, in one launch: depthwise branches -> LDS '
                          '-> MFMA B operand -> second matrix stage on the accumulators; %d launches/forward, the dominant kernel of a pass)' % n_exp,
                'covered_shapes': 'the fused launch exists for (n, columns) = (128, 30 | 32 | 64) and (64, 60 | 64 | 128): 480-, 512- and 1024-pixel-wide inputs, any height; '
                                  'other widths take K1 / K2 / K3 as three launches with identical results (tests/test_gpu_parity.py::test_model_wide_inputs_fused_and_fallback); '
                                  'this figure is the 480-wide case, the 1024-wide one is the cityscapes_512x1024 field',
                'traffic': fam_traffic.get('exp'), 'traffic_source': fam_traffic.get('source'),
                'accounting_note': family_note + '.  achieved = the matrix FLOPs (expansion 2 * 4n * n per pixel, + 2 * n * n where the next projection is fused in) / launch time; the depthwise '
                                   'vector work (36 FMAs per reduced channel and pixel) runs on the same SIMD issue and is not counted; '
                                   'hbm_view = 4 * 9n * H * W (+ 4 * n * H * W) bytes per image / launch time',
                'timing': ('HIP event pair around each launch of one eager pass at batch 16 (stream parked behind a spin kernel; the '
                           'pair\'s own overhead, calibrated per shape on the isolated kernel, subtracted; median of 7 passes); '
                           'rocprofv3 in the same kind of pass: rocprof_avg_launch_us' if 'exp' in ip1 else
                           'isolated re-issues (in-pass measurement failed: %s)' % ip1.get('error')),
                'event_pair_overhead_us': ip1.get('event_pair_overhead_us'),
                'rocprof_avg_launch_us': rocprof_us.get('exp'), 'rocprof_source': rocprof_src}),
            'roofline_k2': dict(roof_k2(ip1 if 'k2' in ip1 else iso) or {}, **{
                'kernel': 'eesp_dw_hff (standalone K2: eesp_dw_stream2_kernel + eesp_dw_direct_kernel, the strided blocks)',
                'traffic': fam_traffic.get('k2'), 'traffic_source': fam_traffic.get('source'),
                'rocprof_avg_launch_us': rocprof_us.get('k2'), 'rocprof_source': rocprof_src}),
            'roofline_family': None if 'all' not in ip1 else {
                'kernel': 'all %d EESP depthwise launches of a forward (fused + standalone) against their algorithmic bytes' % k2_launches,
                'bound': 'hbm', 'achieved': ip1['all']['gbs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(ip1['all']['gbs'] / HBM_PEAK_GBS, 4), 'avg_launch_us': ip1['all']['avg_launch_us'],
                'algorithmic_bytes_per_launch': ip1['all']['bytes_per_launch'],
                'note': 'rounds 1-3: 13 standalone K2 launches, 34.4 MB and 14.2 us per launch = 0.30; the concatenation K2 wrote and K3 read '
                        'back (4 * 8n * H * W bytes per image and block) no longer exists'},
            'roofline_isolated': {'kernel': 'the same launches, each re-issued 20x back to back on warm tensors (best of 3)',
                                  'fused': roof_exp(iso), 'k2': roof_k2(iso)},
            'roofline_bs%d' % (BATCH * group): None if (group == 1 or 'exp' not in in_pass.get(group, {})) else {
                'kernel': 'in-pass at the batch the lanes launch (%d consecutive batches of 16 per launch)' % group,
                'fused': roof_exp(in_pass[group]), 'k2': roof_k2(in_pass[group])},
            'roofline_bs64': None if not k2_ms64 else {
                'kernel': 'standalone K2, same shapes at batch 64 (isolated)',
                'achieved': round(4 * sum(call_bytes(c) for c in calls if c[0] == 'k2') / (sum(k2_ms64) * 1e-3) / 1e9, 1),
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(4 * sum(call_bytes(c) for c in calls if c[0] == 'k2') / (sum(k2_ms64) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                'avg_launch_us': round(sum(k2_ms64) / len(k2_ms64) * 1e3, 3)},
            # the whole hot path against SURVEY section 8(d)'s algorithmic activation traffic (356.9 MB/image at 288x480,
            # convs as in+out, the EESP branches as one shared read, BN/PReLU/add/cat fused = 0)
            'single_in_flight': single,
            'path_roofline': {'algorithmic_bytes_per_image': PATH_BYTES_PER_IMAGE, 'traffic_bytes_per_image': path_traffic,
                              'traffic_source': path_traffic_src,
                              'achieved': round(PATH_BYTES_PER_IMAGE * BATCH * args.steps / elapsed / 1e9, 1),
                              'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                              'frac': round(PATH_BYTES_PER_IMAGE * BATCH * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 4)},
        }
        pk = os.path.join(ROOT, 'profiles', newest(['r05_per_kernel.json', 'r04_per_kernel.json', 'r03_per_kernel.json']) or 'none')
        if os.path.exists(pk):
            # per-kernel table of one label pass (us, MB, fraction of the HBM roof), from a rocprofv3 --kernel-trace --stats run of
            # `bench.py --profile-pass --in-flight 1` (no K2 re-issues in it) + the PMC traffic passes; tools/per_kernel.py
            out['per_kernel'] = json.load(open(pk))
        out.update(multi)
        def extra(name, fn, *a_):
            # an extra field that fails is reported inside the line; it never costs the headline
            try:
                out[name] = fn(*a_)
            except Exception as e_:      # noqa: BLE001
                out[name] = {'error': repr(e_)[:300]}
        if world == 1 and not args.no_three_source:
            extra('three_source', three_source_rate, dev)
            extra('cityscapes_512x1024', cityscapes_rate, dev)
        if world == 1 and not args.no_train:
            extra('train_step', train_step_rate, dev)
            extra('eval_step', eval_step_rate, dev)
            extra('supervised_step', supervised_step_rate, dev)
        if world == 1 and not args.no_aspp:
            extra('aspp_head', aspp_head_rate, dev)
        if world == 1 and not args.no_io:
            extra('loader_io', loader_io_rate, dev)
        if world == 1 and not args.no_cpu_baseline:
            extra('cpu_baseline', cpu_baseline, sd, shape)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
