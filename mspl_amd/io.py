"""The steps on either side of the hot path: batched image/label input transforms on the device, the
pseudo-label PNG writer, and the `tgt_train.lst` round trip.

Reference surface mirrored (paths relative to the reference root):
  data_loader/segmentation/greenhouse.py:152-270           GreenhouseRGBDSegmentation: list parsing, PIL decode, transforms
  transforms/segmentation/data_transforms.py:15-66,191-212 Tensorize / Normalize / RandomFlip / Resize
  uest_seg_multi_os.py:720-728                             update_image_list
  uest_seg_multi_os.py:923-940                             per-image `Image.fromarray(label).save(png)` + path lists

What changes: the reference resizes and normalises one image at a time with Pillow on the host inside the DataLoader
(workers=0, uest_seg_multi_os.py:577) and blocks on a PNG encode per image in the label loop.  Here decoded uint8 images
go to the device as they are (3 B/pixel instead of 12) and `Preprocessor` does Resize + Normalize for the whole batch in two
launches with Pillow's exact fixed-point arithmetic; `LabelWriter` copies the merged uint8 maps back on a side stream into
pinned memory and native worker threads (C++ + zlib inside the library) encode/write the PNGs while the next batch is on the GPU.  Decoding the source JPEG /
PNG files stays with PIL on the host (file formats are outside the path).
"""
import ctypes
import os
import struct
import zlib

import numpy as np
import torch

from ._native import check, lib

def host_cores():
    """Cores this process may run on (the affinity mask where the platform has one: a container's share, not the machine)."""
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def default_writer_workers(world=1):
    """PNG encoding is the host-side limit of the label loop (18 400 images/s with 8 workers against 15 000 from the GPU): most of
    this rank's share of the cores, at most 12."""
    return max(4, min(12, host_cores() // max(1, min(world, 8)) - 2))


MEAN = [0.485, 0.456, 0.406]      # transforms/classification/data_transforms.py:10-11
STD = [0.229, 0.224, 0.225]


# ------------------------------------------------------------------ list files
def update_image_list(tgt_train_lst, image_path_list, label_path_list, depth_path_list=None):
    """uest_seg_multi_os.py:720-728: one `image,label[,depth]` line per image."""
    with open(tgt_train_lst, 'w') as f:
        for idx in range(len(image_path_list)):
            if depth_path_list:
                f.write('%s,%s,%s\n' % (image_path_list[idx], label_path_list[idx], depth_path_list[idx]))
            else:
                f.write('%s,%s\n' % (image_path_list[idx], label_path_list[idx]))


def read_image_list(data_file, use_depth=False, root=None, check_files=True):
    """The list parsing of GreenhouseRGBDSegmentation.__init__ (greenhouse.py:162-197): comma-separated, right-stripped,
    every named file must exist (AssertionError like the reference's `assert os.path.isfile`)."""
    if root:
        data_file = os.path.join(root, data_file)
    images, masks, depths = [], [], []
    with open(data_file, 'r') as lines:
        for line in lines:
            parts = line.split(',')
            rgb, label = parts[0].rstrip(), parts[1].rstrip()
            if check_files:
                assert os.path.isfile(rgb), 'Not found : ' + rgb
                assert os.path.isfile(label), 'Not found : ' + label
            if use_depth:
                depth = parts[2].rstrip()
                if check_files:
                    assert os.path.isfile(depth), 'Not found : ' + depth
                depths.append(depth)
            images.append(rgb)
            masks.append(label)
    return images, masks, depths


# ------------------------------------------------------------------ device-side Resize + Normalize
class Preprocessor(object):
    """`Compose([Resize(size), Normalize() | Tensorize()])` (greenhouse.py:216-222) for a batch, on the device.

    size = (W, H) in PIL order like the reference's `size=(480, 256)`.  Call with uint8 tensors:
    rgb (N,Hs,Ws,3), label (N,Hs,Ws) or None, depth (N,Hs,Ws) or None, flip (N,) bool or None (RandomFlip's coin, drawn by
    the caller).  Host tensors are uploaded (pinned + non_blocking when possible).  Returns
    (rgb fp32 (N,3,H,W), label int64 (N,H,W) | None, depth fp32 (N,1,H,W) | None) on the device, bit-identical to the
    reference's PIL/torchvision pipeline."""

    def __init__(self, size=(480, 256), normalize=True, device='cuda'):
        self.size = size if isinstance(size, tuple) else (size, size)
        self.normalize = normalize
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('mspl_amd: Preprocessor runs on the GPU (no CPU path)')
        self.mean = torch.tensor(MEAN, dtype=torch.float32, device=self.device)
        self.std = torch.tensor(STD, dtype=torch.float32, device=self.device)
        self._tables = {}

    def _bilinear_tables(self, n_in, n_out):
        key = ('b', n_in, n_out)
        if key not in self._tables:
            k = lib.mspl_resample_ksize(n_in, n_out)
            if k <= 0:
                check(k)
            bounds = np.zeros((n_out, 2), np.int32)
            kk = np.zeros((n_out, k), np.int32)
            check(lib.mspl_resample_coeffs(n_in, n_out, bounds.ctypes.data, kk.ctypes.data))
            self._tables[key] = (torch.from_numpy(bounds).to(self.device), torch.from_numpy(kk).to(self.device), k)
        return self._tables[key]

    def _nearest_table(self, n_in, n_out):
        key = ('n', n_in, n_out)
        if key not in self._tables:
            idx = np.zeros(n_out, np.int32)
            check(lib.mspl_nearest_index(n_in, n_out, idx.ctypes.data))
            self._tables[key] = torch.from_numpy(idx).to(self.device)
        return self._tables[key]

    def _up(self, t, name, ndim):
        if t.dtype != torch.uint8 or t.dim() != ndim:
            raise RuntimeError('mspl_amd: %s must be a uint8 tensor with %d dims, got %s %s' % (name, ndim, t.dtype, tuple(t.shape)))
        return t.to(self.device, non_blocking=True).contiguous()

    def _bilinear(self, src, C, mean, std, flip, out=None):
        N, Hs, Ws = src.shape[:3]
        W, H = self.size
        yb, yk, ky = self._bilinear_tables(Hs, H)
        xb = xk = tmp = None
        kx = 0
        if Ws != W:
            xb, xk, kx = self._bilinear_tables(Ws, W)
            tmp = torch.empty((N, Hs, W, C), dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty((N, C, H, W), dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != (N, C, H, W) or out.dtype != torch.float32 or not out.is_cuda or not out.is_contiguous():
            raise RuntimeError('mspl_amd: Preprocessor out= must be a contiguous CUDA float32 tensor of shape %s, got %s %s'
                               % ((N, C, H, W), tuple(out.shape), out.dtype))
        p = lambda t: None if t is None else t.data_ptr()
        check(lib.mspl_preprocess_u8_fwd(src.data_ptr(), N, Hs, Ws, C, H, W, p(xb), p(xk), kx, yb.data_ptr(), yk.data_ptr(), ky,
                                         p(mean), p(std), p(flip), p(tmp), out.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream))
        return out

    def __call__(self, rgb, label=None, depth=None, flip=None, out=None):
        """out: optional destination of the image tensor (N,3,H,W) -- e.g. the static input slot of a captured label pass
        (PipelinedLabelPass.static_inputs()[next_lane]): the transform then writes the network input in place, no copy follows."""
        rgb = self._up(rgb, 'rgb', 4)
        if rgb.shape[3] != 3:
            raise RuntimeError('mspl_amd: rgb must be (N,H,W,3) uint8, got %s' % (tuple(rgb.shape),))
        N = rgb.shape[0]
        if flip is not None:
            flip = torch.as_tensor(flip).to(torch.uint8).to(self.device).contiguous()
            if flip.numel() != N:
                raise RuntimeError('mspl_amd: flip needs one flag per image')
        x = self._bilinear(rgb, 3, self.mean if self.normalize else None, self.std if self.normalize else None, flip, out)
        y = d = None
        if label is not None:
            label = self._up(label, 'label', 3)
            if label.shape[0] != N:
                raise RuntimeError('mspl_amd: label batch %d != image batch %d' % (label.shape[0], N))
            Hs, Ws = label.shape[1:]
            W, H = self.size
            y = torch.empty((N, H, W), dtype=torch.int64, device=self.device)
            check(lib.mspl_resize_label_fwd(label.data_ptr(), N, Hs, Ws, H, W, self._nearest_table(Hs, H).data_ptr(),
                                            self._nearest_table(Ws, W).data_ptr(), None if flip is None else flip.data_ptr(),
                                            y.data_ptr(), torch.cuda.current_stream().cuda_stream))
        if depth is not None:
            depth = self._up(depth, 'depth', 3)
            if depth.shape[0] != N:
                raise RuntimeError('mspl_amd: depth batch %d != image batch %d' % (depth.shape[0], N))
            d = self._bilinear(depth.unsqueeze(3), 1, None, None, flip)
        return x, y, d


# ------------------------------------------------------------------ label PNG writer
_PNG_SIG = b'\x89PNG\r\n\x1a\n'


def _chunk(typ, body):
    return struct.pack('>I', len(body)) + typ + body + struct.pack('>I', zlib.crc32(typ + body) & 0xffffffff)


def encode_png_gray8(arr, level=6):
    """8-bit single-channel PNG of a (H,W) uint8 array: what `Image.fromarray(label.astype(np.uint8)).save(path)` stores
    (uest_seg_multi_os.py:929-931) as far as a decoder can tell (mode 'L', same pixels).  Class-id maps are piecewise
    constant, so the Up filter (row minus the row above) turns them into mostly zeros before deflate."""
    arr = np.ascontiguousarray(arr)
    if arr.dtype != np.uint8 or arr.ndim != 2:
        raise ValueError('encode_png_gray8: expected a (H,W) uint8 array, got %s %s' % (arr.dtype, arr.shape))
    h, w = arr.shape
    raw = np.empty((h, w + 1), np.uint8)
    raw[:, 0] = 2
    raw[0, 1:] = arr[0]
    np.subtract(arr[1:], arr[:-1], out=raw[1:, 1:])              # uint8 wrap-around = PNG's modulo-256 arithmetic
    ihdr = struct.pack('>IIBBBBB', w, h, 8, 0, 0, 0, 0)
    return _PNG_SIG + _chunk(b'IHDR', ihdr) + _chunk(b'IDAT', zlib.compress(raw.tobytes(), level)) + _chunk(b'IEND', b'')


class LabelWriter(object):
    """Asynchronous replacement of the per-image save in the label loop (uest_seg_multi_os.py:923-940).

    `submit(names, labels)` takes the merged (N,H,W) uint8 maps as they leave the label pass (device tensor), starts a
    device->pinned-host copy of a snapshot on a side stream and returns at once (the caller may overwrite `labels`); native
    worker threads (mspl_png_writer_*, host C++ + zlib: no interpreter lock involved) wait for the copy, encode and write
    `<save_dir>/<image_name>.png` (image_name = basename without its extension, :924-926).  `close()` drains the queue and
    returns (image_path_list, label_path_list[, depth_path_list]) in submission order -- the arguments of
    update_image_list.  At most `max_inflight` batches are staged (pinned buffers are reused); beyond that submit() waits
    for the oldest one.  Also usable as a context manager."""

    def __init__(self, save_dir, workers=4, use_depth=False, level=3, max_inflight=None):
        self.save_dir = save_dir
        os.makedirs(save_dir, exist_ok=True)
        self.use_depth = use_depth
        self.level = level
        self.image_paths, self.label_paths, self.depth_paths = [], [], []
        self.max_inflight = max_inflight or 2 * max(1, workers)
        self._handle = lib.mspl_png_writer_create(max(1, workers), level)
        if not self._handle:
            raise RuntimeError('mspl_amd: could not start the PNG writer (workers=%r level=%r)' % (workers, level))
        self._free = {}             # shape -> pinned staging buffers not in use
        self._staged = 0            # staging buffers in existence
        self._inflight = []         # (ticket, staging buffer or None, keep-alive objects) in submission order
        self._stream = None

    def label_path(self, path_name):
        name = path_name.split('/')[-1]
        return '%s/%s.png' % (self.save_dir, name.rsplit('.', 1)[0])

    def _retire(self, block):
        """Collect finished batches (all of them when block is set to 'all', the oldest one when True)."""
        while self._inflight:
            ticket, host, _ = self._inflight[0]
            rc = lib.mspl_png_writer_poll(self._handle, ticket, 1 if block else 0)
            if rc == 0:
                return
            self._inflight.pop(0)
            if host is not None:
                self._free.setdefault(tuple(host.shape), []).append(host)
            if rc < 0:
                check(rc)
            if block is True:
                return

    def _staging(self, shape):
        self._retire(False)
        while True:
            pool = self._free.get(shape)
            if pool:
                return pool.pop()
            if self._staged < self.max_inflight:
                self._staged += 1
                return torch.empty(shape, dtype=torch.uint8, pin_memory=True)
            self._retire(True)                              # back-pressure: wait for the oldest batch in flight

    def warm(self, shape, count=None):
        """Allocate the staging buffers for (N,H,W) batches up front (first-use hipHostMalloc costs ~0.5 ms each)."""
        n = min(count or self.max_inflight, self.max_inflight) - self._staged
        bufs = [self._staging(tuple(shape)) for _ in range(max(0, n))]
        self._free.setdefault(tuple(shape), []).extend(bufs)

    def submit(self, names, labels, stream=None):
        """stream: issue the device -> host copy on THIS stream, straight from `labels` (no snapshot, no side stream).  For callers
        whose `labels` is only ever overwritten by work queued later on that same stream (a PipelinedLabelPass lane's static output
        and the lane's stream): stream order then protects the copy, and the copy needs no hardware queue of its own."""
        if labels.dtype != torch.uint8 or labels.dim() != 3 or labels.shape[0] != len(names):
            raise RuntimeError('mspl_amd: LabelWriter.submit expects (N,H,W) uint8 labels and N names, got %s %s / %d names'
                               % (labels.dtype, tuple(labels.shape), len(names)))
        if self._handle is None:
            raise RuntimeError('mspl_amd: LabelWriter is closed')
        paths = [self.label_path(n) for n in names]
        event, staged = None, None
        if labels.is_cuda and stream is not None:
            host = staged = self._staging(tuple(labels.shape))
            with torch.cuda.stream(stream):
                host.copy_(labels, non_blocking=True)
                event = torch.cuda.Event()
                event.record(stream)
        elif labels.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=labels.device)
            host = staged = self._staging(tuple(labels.shape))
            snap = labels.clone()                          # contents as of submit(): the caller may reuse `labels` at once
            self._stream.wait_stream(torch.cuda.current_stream(labels.device))
            with torch.cuda.stream(self._stream):
                host.copy_(snap, non_blocking=True)
                event = torch.cuda.Event()
                event.record(self._stream)
            snap.record_stream(self._stream)
        else:
            host = labels.contiguous().clone()
        N, H, W = host.shape
        cpaths = (ctypes.c_char_p * N)(*[p.encode() for p in paths])
        ticket = lib.mspl_png_writer_submit(self._handle, host.data_ptr(), N, H, W, cpaths, None if event is None else event.cuda_event)
        if ticket < 0:
            check(int(ticket))
        self._inflight.append((ticket, staged, (host, event)))
        self.image_paths += list(names)
        self.label_paths += paths
        if self.use_depth:
            self.depth_paths += [n.replace('color', 'depth') for n in names]          # uest_seg_multi_os.py:936

    def close(self):
        if self._handle is not None:
            try:
                self._retire('all')
            finally:
                lib.mspl_png_writer_destroy(self._handle)
                self._handle = None
        if self.use_depth:
            return self.image_paths, self.label_paths, self.depth_paths
        return self.image_paths, self.label_paths

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
