#!/usr/bin/env python3
"""Host-side cost of each piece of the frames-in / files-out chain (bench.py loader_io.end_to_end): per-call host time of the
transform, the label-pass submit and the writer submit, and the whole chain with / without writing the transform's output straight
into the lane's static input slot."""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mspl_amd import models, uest
from mspl_amd.io import LabelWriter, Preprocessor
from tests.synth import synth_image_u8, synth_state_dict
B = 16
frames = np.stack([synth_image_u8(360, 480, 900 + i)[0] for i in range(B)])
pinned = torch.from_numpy(frames).pin_memory()
names = ['/d/color/frame_%03d.jpg' % i for i in range(B)]
a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
m = models.ESPDNetwithUncertaintyEstimation(a, classes=13, dataset='camvid', fix_pyr_plane_proj=True)
m.load_state_dict(synth_state_dict(m.state_dict(), 0))
pre = Preprocessor(size=(480, 288))
for inplace in (False, True):
    lp = uest.PipelinedLabelPass(lambda: uest.SelfLabelPass(m, classes=13, device='cuda', use_graph=True, with_kld=False), depth=3, group=2)
    with tempfile.TemporaryDirectory() as d:
        w = LabelWriter(d, workers=int(os.environ.get("W", "8")))
        w.warm((B, 288, 480))
        t_pre = t_lp = t_w = 0.0

        def chain(nb, timed=False):
            global t_pre, t_lp, t_w
            xs = lp.static_inputs((B, 3, 288, 480)) if inplace else None
            for _ in range(nb):
                t0 = time.perf_counter()
                if inplace and xs is not None and xs[lp.next_lane] is not None:
                    x = pre(pinned, out=xs[lp.next_lane])[0]
                else:
                    x = pre(pinned)[0]
                t1 = time.perf_counter()
                r = lp(x)
                t2 = time.perf_counter()
                if r is not None:
                    w.submit(names, r[0])
                t3 = time.perf_counter()
                if timed:
                    t_pre += t1 - t0; t_lp += t2 - t1; t_w += t3 - t2
            for r in lp.flush():
                w.submit(names, r[0])
        chain(12)
        w._retire('all')
        torch.cuda.synchronize()
        nb = int(os.environ.get("NB", "36"))
        t0 = time.perf_counter()
        chain(nb, True)
        torch.cuda.synchronize()
        tg = time.perf_counter() - t0
        w.close()
        ta = time.perf_counter() - t0
    print('inplace=%s: gpu-side %.0f img/s, end-to-end %.0f img/s; host ms per batch: transform %.3f, label submit %.3f, writer submit %.3f'
          % (inplace, nb * B / tg, nb * B / ta, t_pre / nb * 1e3, t_lp / nb * 1e3, t_w / nb * 1e3), flush=True)
    del lp
