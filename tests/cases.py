"""Case tables shared by tests/golden/make_golden.py (reference side) and the tests (oracle / HIP side)."""

# name -> (kind, constructor kwargs, input shape, optional second-input shape)
LAYER_CASES = {
    # EESP stride 1, d = [1,2,3,4] (r_lim 9) and d = [1,1,2,3] (r_lim 7); residual path active
    'eesp_s1_r9': ('eesp', dict(nIn=32, nOut=32, stride=1, k=4, r_lim=9), (2, 32, 20, 28), None),
    'eesp_s1_r7': ('eesp', dict(nIn=64, nOut=64, stride=1, k=4, r_lim=7), (2, 64, 12, 15), None),
    # EESP with nIn != nOut: no residual
    'eesp_s1_nores': ('eesp', dict(nIn=16, nOut=32, stride=1, k=4, r_lim=13), (1, 16, 9, 14), None),
    # DownSampler with image reinforcement (image is 2x the feature resolution, pooled twice)
    'down_r13_reinf': ('down', dict(nin=16, nout=48, k=4, r_lim=13, reinf=True), (2, 16, 24, 40), (2, 3, 48, 80)),
    # DownSampler called without the image (ESPNetv2 level2_0, model/segmentation/espnetv2.py:127)
    'down_r9_noimg': ('down', dict(nin=32, nout=64, k=4, r_lim=9, reinf=True), (1, 32, 16, 20), None),
    # odd input size: 3x3/s2/p1 pooling and strided depthwise on 15x21
    'down_r11_odd': ('down', dict(nin=16, nout=32, k=4, r_lim=11, reinf=True), (1, 16, 15, 21), (1, 3, 60, 84)),
    # EfficientPyrPool, with and without the trailing BR (biased 1x1 in the latter)
    'pyr_br': ('pyr', dict(in_planes=24, proj_planes=8, out_planes=12, last_layer_br=True), (2, 24, 14, 22), None),
    'pyr_nobr': ('pyr', dict(in_planes=12, proj_planes=16, out_planes=5, last_layer_br=False), (1, 12, 16, 30), None),
    # tiny map: every scale < 1 hits the clamp-to-5 branch (efficient_pyramid_pool.py:43-44)
    'pyr_tiny': ('pyr', dict(in_planes=8, proj_planes=4, out_planes=6, last_layer_br=True), (1, 8, 6, 9), None),
    # EfficientPWConv: grouped (gcd 8) and depthwise (gcd 16) expansion
    'pw_g8': ('pw', dict(nin=32, nout=24), (2, 32, 10, 14), None),
    'pw_dw': ('pw', dict(nin=16, nout=16), (1, 16, 12, 9), None),
}

# whole-model cases: name -> (model kind, s, classes, dataset, input shape, sd seed, input seed)
MODEL_CASES = {
    'ue_c13_small': ('espdnetue', 2.0, 13, 'camvid', (1, 3, 48, 64), 11, 0),
    'ue_c5_small': ('espdnetue', 2.0, 5, 'greenhouse', (2, 3, 32, 48), 12, 1),
    'ue_c20_small': ('espdnetue', 2.0, 20, 'city', (1, 3, 32, 32), 13, 2),
    'v2_s05_c13_small': ('espnetv2', 0.5, 13, 'camvid', (2, 3, 32, 64), 14, 3),
    'ue_c13_256x480': ('espdnetue', 2.0, 13, 'camvid', (1, 3, 256, 480), 11, 4),
}

# RGB-D cases (x_d given, espdnet_ue.py:186-270): name -> (classes, dataset, input shape, sd seed, input seed, depth seed,
# dense_fuse, trainable_fusion)
RGBD_CASES = {
    'ue_rgbd_gate': (5, 'greenhouse', (2, 3, 32, 48), 31, 6, 7, False, True),
    'ue_rgbd_dense': (13, 'camvid', (1, 3, 48, 64), 32, 8, 9, True, True),
    'ue_rgbd_add': (5, 'greenhouse', (1, 3, 32, 32), 33, 10, 11, False, False),
}

# single-head ESPDNetSegmentation (model/segmentation/espdnet.py): name -> (classes, dataset, input shape, sd seed, input seed,
# depth seed or None, dense_fuse, trainable_fusion)
ESPDNET_CASES = {
    'espdnet_c5_rgb': (5, 'greenhouse', (2, 3, 32, 48), 41, 12, None, False, True),
    'espdnet_c13_rgbd': (13, 'camvid', (1, 3, 48, 64), 42, 13, 14, False, True),
}

# loader transforms: name -> (source H, source W, PIL size (W,H), seed, normalize, flip, with depth)
IMAGEIO_CASES = {
    'camvid_360x480_to_480x256': (360, 480, (480, 256), 50, True, False, False),      # BASELINE shape: vertical pass only
    'camvid_360x480_to_480x288': (360, 480, (480, 288), 51, True, True, False),
    'big_720x960_to_480x256': (720, 960, (480, 256), 52, True, False, True),           # 2x down-scale: 5-tap windows
    'upscale_100x130_to_480x256': (100, 130, (480, 256), 53, False, False, True),
    'odd_37x53_to_64x48': (37, 53, (64, 48), 54, True, True, True),
    'same_256x480': (256, 480, (480, 256), 55, True, False, False),
}

TRAIN_CASE = dict(s=2.0, classes=5, dataset='greenhouse', shape=(2, 3, 32, 48), sd_seed=21, in_seed=5,
                  lr=5e-4, weight_decay=5e-4, ignore_idx=4)
# golden file -> case.  The second one (round 5) is large enough for the kernels the small one never reaches: the decoder's 48-column
# maps take the streaming pyramid forward / backward kernels, the level-3 / 4 planes (8x12, 4x6 pixels: multiples of 4) the matrix-core
# weight gradients and the fused EESP backward.
TRAIN_CASES = {'train_step': TRAIN_CASE,
               'train_step_64x96': dict(s=2.0, classes=5, dataset='greenhouse', shape=(2, 3, 64, 96), sd_seed=23, in_seed=7,
                                        lr=5e-4, weight_decay=5e-4, ignore_idx=4)}


# one supervised iteration (train_seg_ue + SGD with learning-rate groups, batch-statistics BatchNorm), SURVEY 8f-4
SUPERVISED_CASE = dict(s=2.0, classes=5, dataset='greenhouse', shape=(4, 3, 64, 64), sd_seed=51, in_seed=15, lr=0.009,
                       lr_mult=10.0, momentum=0.9, weight_decay=4e-5, ignore_idx=4, flood=0.015)

# epoch-wise schedules (utilities/lr_scheduler.py): (class name, kwargs, epochs)
LR_CASES = [
    ('CyclicLR', dict(min_lr=0.009, cycle_len=5, steps=[51, 101, 131], gamma=0.5), 160),
    ('CyclicLR', dict(min_lr=0.01, cycle_len=3, steps=[10], gamma=0.7), 40),
    ('FixedMultiStepLR', dict(base_lr=0.1, steps=[30, 60, 90], gamma=0.1), 100),
    ('PolyLR', dict(base_lr=0.009, max_epochs=200, power=0.9), 200),
    ('LinearLR', dict(base_lr=0.009, max_epochs=100), 100),
    ('HybirdLR', dict(base_lr=0.009, clr_max=61, max_epochs=200, cycle_len=5), 200),
    ('HybirdLR', dict(base_lr=0.01, clr_max=11, max_epochs=30, cycle_len=4), 30),
    ('CosineLR', dict(base_lr=0.05, max_epochs=120), 120),
]

# NIDLoss (loss_fns/segmentation_loss.py:54-144): name -> (camera shape, label classes, image_bin, seed)
NID_CASES = {
    'nid_k16_c5': ((2, 3, 32, 48), 5, 16, 80),        # uest: image_bin=args.nid_bin, label_bin=args.classes
    'nid_k32_c13': ((3, 3, 24, 40), 13, 32, 81),      # train_segmentation.py:287: image_bin=32, label_bin=seg_classes
    'nid_k4_c5': ((1, 3, 17, 23), 5, 4, 82),          # fewer image bins than classes: label bins >= K stay empty
}

# ASPP heads (nn_layers/aspp.py, BASELINE configs[4]): name -> (class name, num_classes, input shape, sd seed, input seed).
# Small maps on purpose: the dilations (6 / 12 / 18) then reach past every border.
ASPP_CASES = {
    'aspp_bottleneck_c20': ('ASPP_Bottleneck', 20, (2, 2048, 10, 14), 400, 401),
    'aspp_c13': ('ASPP', 13, (1, 512, 20, 33), 402, 403),
}

# evaluation step (utilities/train_eval_seg.py:249-324 val_seg_ue; uest_seg_multi_os.py:1150-1200 test()): name ->
# (classes, dataset, batch shape, batches, sd seed, first input seed, ignore_idx, class-weight seed, labels may hold 255)
EVAL_CASES = {
    'eval_c5_ign4': (5, 'greenhouse', (2, 3, 64, 96), 3, 61, 600, 4, 62, False),      # the uest target model: class 4 = "other" ignored
    'eval_c13_ign255': (13, 'camvid', (3, 3, 48, 80), 2, 63, 610, 255, 64, True),     # CamVid-style: 255 = void
}

# the two label loops as whole functions (uest_seg_multi_os.py:730-830 generate_pseudo_label, :832-956
# generate_pseudo_label_multi_model): name -> (models [(classes, dataset, os_data, sd seed)], image size (H, W), images, first input
# seed, merge_label_policy, class_weighting).  Image names carry a directory and two dots, so the reference's file-name rule
# (`name.split('/')[-1].rsplit('.', 1)[0]`, :803-805) is exercised.
LABEL_LOOP_CASES = {
    'self_c5': ([(5, 'greenhouse', None, 71)], (48, 64), 7, 700, None, 'normal'),
    'self_c5_unweighted': ([(5, 'greenhouse', None, 72)], (32, 48), 3, 710, None, 'none'),
    'multi3_all': ([(13, 'camvid', 'camvid', 61), (20, 'city', 'cityscapes', 62), (5, 'greenhouse', 'forest', 63)], (48, 64), 5, 720,
                   'all', 'normal'),
    'multi2_half': ([(13, 'camvid', 'camvid', 64), (5, 'greenhouse', 'forest', 65)], (32, 48), 4, 730, 'half', 'normal'),
    # args.eval_training (:749-752, :871-876): models in train() mode under no_grad -> BatchNorm with the statistics of each single image
    'self_c5_evaltrain': ([(5, 'greenhouse', None, 73)], (48, 64), 5, 740, None, 'normal', True),
    'multi2_all_evaltrain': ([(13, 'camvid', 'camvid', 66), (5, 'greenhouse', 'forest', 67)], (48, 64), 3, 750, 'all', 'normal', True),
}
