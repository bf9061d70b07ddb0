"""The functions uest_seg_multi_os.py defines ITSELF, with the reference's own positional signatures.

install_dropin() can only alias what the script imports; `get_output`, `merge_outputs`, `update_image_list`,
`generate_pseudo_label` and `generate_pseudo_label_multi_model` are defined in the script's own namespace
(uest_seg_multi_os.py:669-956) and called from its main() (:500, :527).  `patch_script(globals())`, placed after those
definitions (or `install_dropin(script=globals())`), rebinds the five names to the HIP-backed versions, so main() runs
unchanged:

    tgt_train_lst, class_weights = generate_pseudo_label(model, device, save_path, round_idx,
        tgt_num, label_2_id, valid_labels, args, logger, class_encoding, writer)                  # :527

The adapters read from `args` exactly what the reference functions read (`classes`, `data_tgt_train_list`, `use_traversable`,
`use_depth`, `pin_memory`, `class_weighting`, `merge_label_policy`, `eval_training`, `dataset`) plus three optional fields the
reference does not have: `label_batch_size` (default 16; the reference's loader is batch size 1, :746 -- BatchNorm is in eval mode,
so the maps do not depend on it), `label_in_flight` (3), `label_batches_per_launch` (2).  The arguments the reference functions
accept and never use (`tgt_num`, `label_2_id`, `valid_labels`, `class_encoding`, `writer`; the ScoreUpdater built at :733 is
reset and dropped) are accepted and ignored.
"""
import torch

from . import uest
from .io import update_image_list


def _target_loader(args):
    """uest_seg_multi_os.py:739-746 / :842-849: the target-domain list as a loader of `(image, label[, depth], name, _)` tuples.
    The dataset class is the reference's own (data_loader.segmentation.greenhouse, reached through the drop-in overlay)."""
    if getattr(args, 'dataset', 'greenhouse') != 'greenhouse':
        raise RuntimeError('mspl_amd: the label functions build a loader for --dataset greenhouse only (the reference leaves `ds` '
                           'undefined for anything else, uest_seg_multi_os.py:739-746)')
    from data_loader.segmentation.greenhouse import GreenhouseRGBDSegmentation
    ds = GreenhouseRGBDSegmentation(list_name=args.data_tgt_train_list, train=False,
                                    use_traversable=getattr(args, 'use_traversable', False),
                                    use_depth=getattr(args, 'use_depth', False))
    return torch.utils.data.DataLoader(ds, batch_size=int(getattr(args, 'label_batch_size', 16)), shuffle=False,
                                       pin_memory=bool(getattr(args, 'pin_memory', False)))


def _mode(args):
    """:749-752 / :871-876: `--eval-training` labels with the models in train() mode (BatchNorm with the statistics of each single
    image -- the reference's loader has batch size 1); otherwise eval()."""
    return bool(getattr(args, 'eval_training', False))


def _log(logger, round_idx):
    if logger is not None:
        logger.info('###### Start evaluating target domain train set in round {}! ######'.format(round_idx))      # :777


def generate_pseudo_label(model, device, save_path, round_idx, tgt_num=None, label_2_id=None, valid_labels=None, args=None,
                          logger=None, class_encoding=None, writer=None, testloader=None):
    """uest_seg_multi_os.py:730-830 with its own signature; returns (tgt_train_lst, class_weights float32 on `device`)."""
    eval_training = _mode(args)
    loader = testloader if testloader is not None else _target_loader(args)
    _log(logger, round_idx)
    lst, w = uest.generate_pseudo_label(
        model, loader, save_path, eval_training=eval_training, classes=args.classes, class_weighting=getattr(args, 'class_weighting', 'normal'),
        use_depth=getattr(args, 'use_depth', False), device=device, in_flight=int(getattr(args, 'label_in_flight', 3)),
        batches_per_launch=int(getattr(args, 'label_batches_per_launch', 2)))
    print('class_weights : {}'.format(w.cpu().numpy()))   # :826
    return lst, w.to(device)


def generate_pseudo_label_multi_model(model_list, os_data_list, device, save_path, round_idx, tgt_num=None, label_2_id=None,
                                      valid_labels=None, args=None, logger=None, class_encoding=None, writer=None, testloader=None):
    """uest_seg_multi_os.py:832-956 with its own signature."""
    eval_training = _mode(args)
    loader = testloader if testloader is not None else _target_loader(args)
    _log(logger, round_idx)
    lst, w = uest.generate_pseudo_label_multi_model(
        model_list, os_data_list, loader, save_path, eval_training=eval_training, classes=args.classes,
        merge_label_policy=getattr(args, 'merge_label_policy', 'all'), class_weighting=getattr(args, 'class_weighting', 'normal'),
        use_depth=getattr(args, 'use_depth', False), device=device, in_flight=int(getattr(args, 'label_in_flight', 3)),
        batches_per_launch=int(getattr(args, 'label_batches_per_launch', 1)))
    print('class_weights : {}'.format(w.cpu().numpy()))   # :948
    return lst, w.to(device)


SCRIPT_FUNCTIONS = {
    'get_output': uest.get_output,                                          # :669
    'merge_outputs': uest.merge_outputs,                                    # :695
    'update_image_list': update_image_list,                                 # :720
    'generate_pseudo_label': generate_pseudo_label,                         # :730
    'generate_pseudo_label_multi_model': generate_pseudo_label_multi_model,  # :832
}


def patch_script(namespace):
    """Rebind the script-level functions in `namespace` (the script's globals() or its module object).  Returns the names bound."""
    ns = namespace if isinstance(namespace, dict) else vars(namespace)
    ns.update(SCRIPT_FUNCTIONS)
    return sorted(SCRIPT_FUNCTIONS)
