"""Oracle: uncertainty estimator, label merge, losses and metrics of the pseudo-label pass.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy for the integer work, torch-CPU fp32
for the floating point.  Citations are relative to /root/reference.
"""
import numpy as np
import torch
import torch.nn.functional as F

# data_loader/segmentation/greenhouse.py:15-58 (source-dataset class id -> 5 greenhouse classes).
ID_CAMVID_TO_GREENHOUSE = np.array([4, 2, 2, 3, 3, 1, 2, 2, 2, 4, 4, 2, 4])
ID_CITYSCAPES_TO_GREENHOUSE = np.array([3, 3, 2, 2, 2, 2, 2, 2, 1, 3, 4, 4, 4, 2, 2, 2, 2, 2, 2, 4])
ID_FOREST_TO_GREENHOUSE = np.array([3, 1, 1, 2, 2])
LUTS = {'camvid': ID_CAMVID_TO_GREENHOUSE, 'cityscapes': ID_CITYSCAPES_TO_GREENHOUSE,
        'forest': ID_FOREST_TO_GREENHOUSE}
GREENHOUSE_CLASSES = 5
NO_AGREEMENT_CLASS = 4  # the literal 4 written at uest_seg_multi_os.py:716


def pixelwise_kld(d1, d2):
    """PixelwiseKLD.forward, loss_fns/segmentation_loss.py:181-189: sum_c p1*(logp1 - logp2)."""
    p1 = F.softmax(d1, dim=1)
    lp1 = F.log_softmax(d1, dim=1)
    lp2 = F.log_softmax(d2, dim=1)
    return torch.sum(p1 * lp1 - p1 * lp2, dim=1)


def get_output(pred, pred_aux):
    """get_output, uest_seg_multi_os.py:685-693, applied to every batch element.

    The reference keeps batch element 0 only (bs=1 loop); the batched restatement returns the same
    thing for each element.  Returns (softmax(pred + 0.5 aux) as (N,C,H,W), kld as (N,H,W)).
    """
    prob = F.softmax(pred + 0.5 * pred_aux, dim=1)
    return prob, pixelwise_kld(pred, pred_aux)


def argmax_labels(prob):
    """uest_seg_multi_os.py:903-904: np.argmax over the class axis (first max wins) -> uint8."""
    out = prob.numpy() if isinstance(prob, torch.Tensor) else prob
    return np.asarray(np.argmax(out.transpose(0, 2, 3, 1), axis=3), dtype=np.uint8)


def to_greenhouse(amax, os_data):
    """uest_seg_multi_os.py:907-912: LUT for the three outsource datasets, identity otherwise."""
    lut = LUTS.get(os_data)
    return amax if lut is None else lut[amax]


def merge_outputs(amax_outputs, seg_classes=GREENHOUSE_CLASSES, thresh=None):
    """merge_outputs, uest_seg_multi_os.py:695-718.  amax_outputs: (S, ...) integer class maps."""
    num_data = amax_outputs.shape[0]
    if thresh is None or thresh == 'half':
        thresh = num_data // 2 + 1
    elif thresh == 'all':
        thresh = num_data
    elif isinstance(thresh, int) and not isinstance(thresh, bool) and thresh <= num_data:
        pass
    else:
        thresh = num_data // 2 + 1
    counts = np.array([(amax_outputs == c).sum(axis=0) for c in range(seg_classes)])
    amax = counts.argmax(axis=0)
    amax[counts.max(axis=0) < thresh] = NO_AGREEMENT_CLASS
    return amax


def class_histogram(merged, classes=GREENHOUSE_CLASSES):
    """uest_seg_multi_os.py:919-921."""
    return np.array([(merged == i).sum() for i in range(classes)], dtype=np.float64)


def class_weights_from_histogram(class_array, policy='normal'):
    """uest_seg_multi_os.py:942-947."""
    class_array = np.asarray(class_array, dtype=np.float64)
    if policy == 'normal':
        freq = class_array / class_array.sum()
        w = 1.0 / (freq + 1e-10)
        w[0] = 0.0
        return w
    return np.ones(len(class_array))


def generate_pseudo_label(forward, loader, classes, save_pred_path, class_weighting='normal', use_depth=False):
    """uest_seg_multi_os.py:783-828, the single-model relabelling loop, image by image like the reference (its loader is batch size
    1, :746; a larger batch is walked element by element here): get_output -> argmax (:798) -> class_array (:800-801) -> path lists
    (:803-816) -> class weights (:822-828).  forward(image (1,3,H,W)) -> (pred, pred_aux) logits at full resolution.  Returns
    (image_path_list, label_path_list, depth_path_list, [uint8 (H,W) maps], class_weights float64) -- the PNG files the reference
    writes at :811 hold exactly those maps."""
    image_paths, label_paths, depth_paths, maps = [], [], [], []
    class_array = np.zeros(classes)
    for batch in loader:
        image, names = batch[0], batch[-2]
        for k in range(image.shape[0]):
            main, aux = forward(image[k:k + 1])
            prob, _ = get_output(main, aux)
            amax = np.asarray(np.argmax(prob[0].numpy().transpose(1, 2, 0), axis=2), dtype=np.uint8)      # [0]: :688
            for i in range(classes):
                class_array[i] += (amax == i).sum()
            path_name = names[k]
            image_name = path_name.split('/')[-1].rsplit('.', 1)[0]
            maps.append(amax)
            image_paths.append(path_name)
            label_paths.append('%s/%s.png' % (save_pred_path, image_name))
            if use_depth:
                depth_paths.append(path_name.replace('color', 'depth'))
    return image_paths, label_paths, depth_paths, maps, class_weights_from_histogram(class_array, class_weighting)


def generate_pseudo_label_multi_model(forwards, os_data_list, loader, classes, save_pred_path, merge_label_policy=None,
                                      class_weighting='normal', use_depth=False):
    """uest_seg_multi_os.py:891-950, the multi-source relabelling loop, image by image: per source model get_output -> argmax
    (:901-904) -> LUT (:907-912); merge_outputs with args.merge_label_policy (:916-917) -> class_array (:920-921) -> file-name rule and
    path lists (:923-936) -> class weights (:940-947).  forwards: one callable per source model, image (1,3,H,W) -> (pred, pred_aux).
    Same return value as generate_pseudo_label."""
    image_paths, label_paths, depth_paths, maps = [], [], [], []
    class_array = np.zeros(classes)
    for batch in loader:
        image, names = batch[0], batch[-2]
        for k in range(image.shape[0]):
            srcs = []
            for forward, os_data in zip(forwards, os_data_list):
                main, aux = forward(image[k:k + 1])
                prob, _ = get_output(main, aux)
                amax = np.asarray(np.argmax(prob[0].numpy().transpose(1, 2, 0), axis=2), dtype=np.uint8)
                srcs.append(to_greenhouse(amax, os_data))
            merged = merge_outputs(np.array(srcs), classes, merge_label_policy)
            for i in range(classes):
                class_array[i] += (merged == i).sum()
            path_name = names[k]
            image_name = path_name.split('/')[-1].rsplit('.', 1)[0]
            maps.append(merged.astype(np.uint8))
            image_paths.append(path_name)
            label_paths.append('%s/%s.png' % (save_pred_path, image_name))
            if use_depth:
                depth_paths.append(path_name.replace('color', 'depth'))
    return image_paths, label_paths, depth_paths, maps, class_weights_from_histogram(class_array, class_weighting)


def uw_seg_loss(pred, target, u_weight, class_weights, ignore_idx=None):
    """UncertaintyWeightedSegmentationLoss.forward, loss_fns/segmentation_loss.py:155-175.

    class_weights[ignore_idx] is zeroed (:152-153, on a copy here); the mean runs over ALL pixels.
    """
    cw = class_weights.clone()
    if ignore_idx is not None:
        cw[ignore_idx] = 0.0
    n, c, h, w = pred.shape
    logp = -F.log_softmax(pred, dim=1)
    logp = logp * cw.reshape(1, c, 1, 1)
    logp = logp.gather(1, target.view(n, 1, h, w))
    logp = logp * torch.exp(-u_weight.reshape(n, 1, h, w))
    return logp.mean()


def uest_train_loss(pred, pred_aux, labels, class_weights, ignore_idx=None):
    """uest_seg_multi_os.py:1020-1023: criterion(pred+0.5aux, labels, kld)*20 + kld.mean() (kld not detached)."""
    kld = pixelwise_kld(pred, pred_aux)
    return uw_seg_loss(pred + 0.5 * pred_aux, labels, kld, class_weights, ignore_idx) * 20 + kld.mean()


def miou_areas(output, target, num_classes):
    """MIOU.get_iou, utilities/metrics/segmentation_miou.py:13-44 (uint8 arithmetic, 255 wraps to 0)."""
    pred = output.argmax(1) if output.dim() == 4 else output
    pred = pred.to(torch.uint8) + 1
    target = target.to(torch.uint8) + 1
    pred = pred * (target > 0)
    inter = pred * (pred == target)
    k = num_classes
    a_i = torch.histc(inter.float(), bins=k, min=1, max=k)
    a_p = torch.histc(pred.float(), bins=k, min=1, max=k)
    a_m = torch.histc(target.float(), bins=k, min=1, max=k)
    return a_i.numpy(), (a_p + a_m - a_i + 1e-6).numpy()


def soft_arg_max(A, beta=500.0, epsilon=1e-12):
    """SoftArgMax.soft_arg_max, loss_fns/segmentation_loss.py:124-141 -> (B,1,H,W)."""
    A_max = torch.max(A, dim=1, keepdim=True)[0]
    A_exp = torch.exp((A - A_max) * beta)
    A_softmax = A_exp / (torch.sum(A_exp, dim=1, keepdim=True) + epsilon)
    indices = torch.arange(start=0, end=A.size(1)).float().reshape(1, A.size(1), 1, 1)
    return F.conv2d(A_softmax, indices)


def nid_loss(camera, label, image_bin=16, label_bin=4, bw_camera=0.005, bw_label=0.001, eps=1e-7):
    """NIDLoss.forward, loss_fns/segmentation_loss.py:54-121 (device moves dropped)."""
    K, C = image_bin, label_bin
    cam = torch.sum(camera, 1) / 3
    lab = soft_arg_max(label)
    lab = lab.reshape(lab.size(0), lab.size(2), lab.size(3))
    num_pixel = cam.size(1) * cam.size(2)
    batch = cam.size(0)
    cam_1d = cam.reshape(batch, -1)
    lab_1d = lab.reshape(batch, -1)
    L_c, L_l = 1 / K, 1
    P_c, P_l = [], [torch.zeros(num_pixel) for _ in range(C)]
    for k in range(K):
        mu = L_c * (k + 1 / 2)
        P_c.append(torch.sum(torch.sigmoid((cam_1d - mu + L_c / 2) / bw_camera) - torch.sigmoid((cam_1d - mu - L_c / 2) / bw_camera), 0))
        if k < C:
            P_l[k] = torch.sum(torch.sigmoid((lab_1d - k + L_l / 2) / bw_label) - torch.sigmoid((lab_1d - k - L_l / 2) / bw_label), 0)
    P_c, P_l = torch.stack(P_c), torch.stack(P_l)
    norm = num_pixel * batch
    p_cl, p_c, p_l = torch.mm(P_c, P_l.t()) / norm, torch.sum(P_c, 1) / norm, torch.sum(P_l, 1) / norm
    p_cl, p_c, p_l = p_cl / p_cl.sum(), (p_c / p_c.sum()).reshape(-1, 1), (p_l / p_l.sum()).reshape(-1, 1)
    I = torch.sum(p_cl * (torch.log(p_cl + eps) - torch.log(torch.mm(p_c, p_l.t()) + eps)))
    H = -torch.sum(p_cl * torch.log(p_cl + eps))
    return ((1 - I / H) - 0.95) * 20


def val_seg_ue(forward, loader, class_weights, ignore_idx, num_classes, aux_weight=0.5):
    """val_seg_ue, utilities/train_eval_seg.py:249-324 (aux_weight 0.5), and the body of test(), uest_seg_multi_os.py:1150-1200
    (aux_weight 0: criterion and MIOU on the main head alone).  forward(x) -> (main, aux) full-resolution logits.
    Returns (iou float64[K], average loss) exactly as the reference's meters accumulate them."""
    K = num_classes - 1
    inter_sum, union_sum = np.zeros(K, np.float64), np.zeros(K, np.float64)
    loss_sum, count = 0.0, 0
    crit = torch.nn.CrossEntropyLoss(ignore_index=ignore_idx, weight=class_weights)
    with torch.no_grad():
        for x, y in loader:
            main, aux = forward(x)
            out = main + aux_weight * aux if aux_weight != 0.0 else main
            loss = crit(out, y).mean()
            inter, union = miou_areas(out, y, K)
            inter_sum += inter
            union_sum += union
            loss_sum += float(loss.item()) * x.size(0)
            count += x.size(0)
    return inter_sum / (union_sum + 1e-10), loss_sum / count
