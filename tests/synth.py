"""Deterministic synthetic state dicts and inputs shared by the golden generator and the tests.

Every tensor is a pure function of (key name, shape, seed), so the reference-side generator
(tests/golden/make_golden.py, run once in the build container) and the tests (run anywhere) see
bit-identical weights without committing megabytes of parameters.  BatchNorm statistics, affine
terms and PReLU slopes are randomised on purpose: the reference's own init (gamma=1, beta=0,
mean=0, var=1, slope=0.25) would make a folded-BN bug invisible.
"""
import zlib

import torch


def _gen(key, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)
    return g


def synth_tensor(key, shape, seed, template_keys):
    shape = tuple(shape)
    g = _gen(key, seed)
    stem, _, leaf = key.rpartition('.')
    if leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == 'running_mean':
        return torch.randn(shape, generator=g) * 0.1
    if leaf == 'running_var':
        return torch.rand(shape, generator=g) + 0.5
    is_bn = (stem + '.running_mean') in template_keys
    if len(shape) == 4:  # conv weight, unit-gain so activations stay O(1) through the net
        fan_in = shape[1] * shape[2] * shape[3]
        return torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5
    if leaf == 'weight' and is_bn:
        return torch.rand(shape, generator=g) + 0.5
    if leaf == 'bias':
        return torch.randn(shape, generator=g) * 0.1
    if leaf == 'weight' and len(shape) == 1:  # PReLU slope
        return torch.rand(shape, generator=g) * 0.4 + 0.05
    raise ValueError('unclassified state-dict entry %s %s' % (key, shape))


def synth_state_dict(template, seed):
    """template: mapping key -> tensor or shape (only keys and shapes are used)."""
    keys = set(template.keys())
    out = {}
    for k, v in template.items():
        shape = tuple(v.shape) if hasattr(v, 'shape') else tuple(v)
        out[k] = synth_tensor(k, shape, seed, keys)
    return out


def synth_input(shape, seed):
    g = torch.Generator()
    g.manual_seed(1000003 + seed)
    return torch.randn(tuple(shape), generator=g)


def synth_labels(shape, num_classes, seed):
    g = torch.Generator()
    g.manual_seed(2000003 + seed)
    return torch.randint(0, num_classes, tuple(shape), generator=g)


def synth_image_u8(h, w, seed):
    """(rgb (h,w,3), label (h,w) with ids 0..4 and some 255, depth (h,w)) uint8 numpy arrays: smooth blobs plus noise, so
    that resampling sees both gradients and edges."""
    import numpy as np
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (127 + 90 * np.sin(yy / 9.0 + seed) * np.cos(xx / 13.0)).astype(np.int32)
    rgb = np.clip(base[:, :, None] + rng.integers(-60, 61, (h, w, 3)), 0, 255).astype(np.uint8)
    label = ((yy // 11 + xx // 17 + seed) % 5).astype(np.uint8)
    label[rng.random((h, w)) < 0.02] = 255
    depth = np.clip(base + rng.integers(-30, 31, (h, w)), 0, 255).astype(np.uint8)
    return rgb, label, depth


def synth_nid_inputs(shape, classes, seed):
    """(camera (B,3,H,W) in roughly [-0.3, 1.3], label logits (B,C,H,W)).  About 15 % of the pixels get two ADJACENT classes
    as near-ties (difference of a few 1e-3), so that the soft-arg-max lands between two label bins and the loss has a
    gradient there (with beta = 500 and bw_label = 1e-3 it is zero to fp32 precision everywhere else)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    B, _, H, W = shape
    cam = torch.rand((B, 3, H, W), generator=g) * 1.6 - 0.3
    lab = torch.randn((B, classes, H, W), generator=g) * 3
    tie = torch.rand((B, H, W), generator=g) < 0.15
    c0 = torch.randint(0, classes - 1, (B, H, W), generator=g)
    delta = (torch.rand((B, H, W), generator=g) - 0.5) * 0.008
    top = lab.max(1)[0] + 1.0
    for c in range(classes - 1):
        m = tie & (c0 == c)
        lab[:, c][m] = top[m]
        lab[:, c + 1][m] = (top + delta)[m]
    return cam, lab


def synth_eval_batches(case):
    """The seeded loader of a tests.cases.EVAL_CASES entry: [(images, labels)] -- shared by the golden generator (which feeds it to
    the reference's val_seg_ue) and the tests."""
    C, ds, shape, nb, sd_seed, in_seed, ign, cw_seed, with_void = case
    out = []
    for b in range(nb):
        x = synth_input(shape, in_seed + b)
        y = synth_labels((shape[0],) + shape[2:], C, in_seed + b)
        if with_void:
            void = synth_labels((shape[0],) + shape[2:], 10, in_seed + 50 + b) == 0      # ~10 % void pixels
            y = torch.where(void, torch.full_like(y, 255), y)
        out.append((x, y))
    return out


def grad_sample_index(numel):
    """Indices of the strided gradient sample kept in tests/golden/train_step.npz (shared by generator and test)."""
    step = max(1, numel // 64)
    idx = list(range(0, numel, step))
    if idx[-1] != numel - 1:
        idx.append(numel - 1)
    return torch.tensor(idx, dtype=torch.int64)


def synth_label_loop_images(case):
    """The seeded target-domain images of a tests.cases.LABEL_LOOP_CASES entry: [(image (3,H,W), name)] -- shared by the golden
    generator (which serves them to the reference's label loops through a stub dataset) and the tests."""
    specs, (H, W), n, in_seed, policy, weighting = case[:6]
    return [(synth_input((3, H, W), in_seed + i), '/data/greenhouse/color/seq.%d/frame_%03d.v2.jpg' % (i % 2, i)) for i in range(n)]


def synth_adversarial_logits(C, seed, pixels=2048):
    """(pred, aux) of shape (pixels/64, C, 8, 8) whose combined logits z = pred + 0.5 aux have their two LARGEST entries 0..4 ulp
    apart at every pixel, over four magnitude bands (|z| ~ 0.02, 0.3, 2, 12) -- the inputs on which `np.argmax(softmax(z))`
    (uest_seg_multi_os.py:687-691,798) and an argmax of z itself can differ: fp32 softmax may give both candidates the same
    probability and the first maximum then picks the lower class id.  Returns (pred, aux, a, b, k): classes a != b hold the top two,
    z[b] = z[a] advanced by k ulp (k may be negative).  aux is zero on even pixels and a coarse dyadic value on odd ones (0.5 aux is
    exact either way; pred is chosen so that the SUM has the wanted spacing)."""
    import numpy as np
    rng = np.random.default_rng(4000 + seed)
    n = pixels
    scale = np.array([0.02, 0.3, 2.0, 12.0], dtype=np.float32)[rng.integers(0, 4, n)]
    z = (rng.standard_normal((n, C)).astype(np.float32) * scale[:, None] * np.float32(0.5)).astype(np.float32)
    a = rng.integers(0, C, n)
    b = (a + rng.integers(1, C, n)) % C
    k = rng.integers(-4, 5, n)
    top = (np.abs(z).max(axis=1) + scale * rng.uniform(0.05, 1.0, n).astype(np.float32)).astype(np.float32)
    zb = top.copy()
    for step in range(1, 5):
        up = k >= step
        dn = k <= -step
        zb[up] = np.nextafter(zb[up], np.float32(np.inf))
        zb[dn] = np.nextafter(zb[dn], np.float32(-np.inf))
    z[np.arange(n), a] = top
    z[np.arange(n), b] = zb
    aux = np.zeros((n, C), dtype=np.float32)
    odd = np.arange(n) % 2 == 1
    aux[odd] = (rng.integers(-8, 9, (int(odd.sum()), C)) * 0.25).astype(np.float32)
    pred = (z - np.float32(0.5) * aux).astype(np.float32)
    # the sum the networks' consumers form; where the subtraction above was inexact the spacing is re-imposed on the sum
    zs = (pred + np.float32(0.5) * aux).astype(np.float32)
    bad = (zs[np.arange(n), a] != top) | (zs[np.arange(n), b] != zb) | (np.argsort(zs, axis=1)[:, -1:] != np.where(k > 0, b, a)[:, None]).ravel() & (k != 0)
    aux[bad] = 0.0
    pred[bad] = z[bad]
    shape = (n // 64, 8, 8, C)
    to = lambda t: torch.from_numpy(np.ascontiguousarray(t.reshape(shape).transpose(0, 3, 1, 2)))
    return to(pred), to(aux), a.reshape(shape[:3]), b.reshape(shape[:3]), k.reshape(shape[:3])


def assert_weights_close_after_adam(got, want, lr, steps, tight=5e-5, frac=0.01):
    """Weights of two runs of the same Adam steps whose gradients differ by rounding (float atomics land in a different order).
    Adam divides by sqrt(v): an element whose gradient is rounding noise around zero moves by ~lr per step in the direction of
    that noise, and an ulp of difference in a weight can put one PReLU input of a tiny map on the other side of zero, which moves
    the gradients of that block by per cent (tools/diag_sinks.py).  So: every element within 2 * lr * steps, and all but `frac`
    of them within `tight` -- a missing, doubled or misplaced gradient moves far more than one per cent of the elements.
    got / want: dicts of tensors (state_dicts), tensors or arrays."""
    import numpy as np
    import torch

    def flat(v):
        if isinstance(v, dict):
            return np.concatenate([t.detach().float().cpu().numpy().ravel() for t in v.values() if torch.is_floating_point(t)])
        return (v.detach().float().cpu().numpy() if torch.is_tensor(v) else np.asarray(v, dtype=np.float32)).ravel()
    a, b = flat(got), flat(want)
    assert a.shape == b.shape
    d = np.abs(a - b)
    assert float(d.max()) <= 2.0 * lr * steps + tight, 'largest difference %.3g > 2 * lr * steps' % float(d.max())
    loose = float((d > tight).mean())
    assert loose <= frac, '%.4f of the elements differ by more than %g' % (loose, tight)
