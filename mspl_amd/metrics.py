"""utilities/metrics/segmentation_miou.py on the device: `MIOU(num_classes).get_iou(output, target)`.

The reference moves the argmax and the labels to the CPU every training step for three `torch.histc` calls
(uest_seg_multi_os.py:1032,1198).  Here one integer kernel produces the three histograms; `get_iou` keeps the reference's
return type (two numpy float32 arrays, which costs a 3*K-value copy), `get_iou_device` returns device tensors and no sync.
"""
import torch

from ._native import check, lib
from .ops import _p, _stream


class MIOU(object):
    def __init__(self, num_classes=21):
        self.num_classes = num_classes
        self.epsilon = 1e-6

    def areas(self, output, target):
        """int64 tensor (3, K) on the device: area_inter, area_pred, area_mask."""
        if isinstance(output, tuple):
            output = output[0]
        if not output.is_cuda or not target.is_cuda:
            raise RuntimeError('mspl_amd: MIOU needs CUDA tensors (there is no CPU path)')
        K = int(self.num_classes)
        target = target.to(torch.int64).contiguous()
        hist = torch.zeros((3, K), device=output.device, dtype=torch.int64)
        if output.dim() == 4:                     # raw outputs: argmax inside the kernel
            if output.dtype != torch.float32:
                raise RuntimeError('mspl_amd: MIOU logits must be float32, got %s' % output.dtype)
            output = output.contiguous()
            N, C, H, W = output.shape
            if target.numel() != N * H * W:
                raise RuntimeError('mspl_amd: MIOU target %s does not match output %s' % (tuple(target.shape), tuple(output.shape)))
            check(lib.mspl_miou_areas_fwd(_p(output), None, _p(target), N, C, H * W, K, _p(hist), _stream()))
        else:                                     # already an argmax map
            lab = output.to(torch.uint8).contiguous()
            if target.numel() != lab.numel():
                raise RuntimeError('mspl_amd: MIOU target %s does not match labels %s' % (tuple(target.shape), tuple(lab.shape)))
            check(lib.mspl_miou_areas_fwd(None, _p(lab), _p(target), 1, 0, lab.numel(), K, _p(hist), _stream()))
        return hist

    def get_iou_device(self, output, target):
        h = self.areas(output, target).to(torch.float32)
        return h[0], h[1] + h[2] - h[0] + self.epsilon

    def get_iou(self, output, target):
        inter, union = self.get_iou_device(output, target)
        return inter.cpu().numpy(), union.cpu().numpy()
