"""Oracle: one uest self-training step (forward, UW-loss*20 + KLD, backward, Adam) on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows uest_seg_multi_os.py:1012-1041 with
torch.optim.Adam semantics (L2 weight decay folded into the gradient; parameters whose gradient
is None are skipped entirely).  BatchNorm is frozen (eval mode, Appendix B-3).
"""
import torch

from . import labels as olab
from . import net as onet


def train_step(sd, param_names, x, labels, class_weights, ignore_idx, lr=5e-4, weight_decay=5e-4,
               betas=(0.9, 0.999), eps=1e-8):
    """Returns (loss, {name: grad or None}, {name: updated tensor}) for the first Adam step."""
    work = dict(sd)
    params = {}
    for n in param_names:
        params[n] = sd[n].clone().requires_grad_(True)
        work[n] = params[n]
    main, aux = onet.espdnet_ue_forward(work, x)
    loss = olab.uest_train_loss(main, aux, labels, class_weights, ignore_idx)
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    gmap, new = {}, {}
    b1, b2 = betas
    for (n, p), g in zip(params.items(), grads):
        gmap[n] = g
        if g is None:
            new[n] = p.detach()
            continue
        g = g + weight_decay * p.detach()
        m = (1 - b1) * g
        v = (1 - b2) * g * g
        denom = (v.sqrt() / (1 - b2) ** 0.5) + eps
        new[n] = p.detach() - (lr / (1 - b1)) * (m / denom)
    return loss.detach(), gmap, new


def supervised_step(sd, groups, x, labels, class_weights, ignore_idx, momentum=0.9, weight_decay=4e-5, b=0.015, x_d=None):
    """One iteration of train_seg_ue (utilities/train_eval_seg.py:179-225) with torch.optim.SGD over learning-rate groups
    (train_segmentation.py:244-253): model.train() -> batch-statistics BatchNorm, loss = CrossEntropy(main + 0.5*aux)
    (ignore_index, class weights), flooding (loss-b).abs()+b, first SGD step (momentum buffer = g + wd*p).

    groups: list of (parameter names, lr).  Returns (loss, {name: grad or None}, {name: new value}, updated state dict
    (running statistics after the forward))."""
    work = {k: (v.clone() if k.endswith(('running_mean', 'running_var')) else v) for k, v in sd.items()}
    params = {}
    for names, _ in groups:
        for n in names:
            params[n] = sd[n].clone().requires_grad_(True)
            work[n] = params[n]
    with onet.bn_training():
        main, aux = onet.espdnet_ue_forward(work, x, x_d)
    out = main + 0.5 * aux
    w = None if class_weights is None else class_weights.float()
    loss = torch.nn.functional.cross_entropy(out, labels, weight=w, ignore_index=ignore_idx).mean()
    loss = (loss - b).abs() + b
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    gmap, new = dict(zip(params.keys(), grads)), {}
    for names, lr in groups:
        for n in names:
            g, p = gmap[n], params[n].detach()
            new[n] = p if g is None else p - lr * (g + weight_decay * p)
    return loss.detach(), gmap, new, {k: v.detach() for k, v in work.items()}
