"""Model presets with the hyper-parameters of the reference's live configs, for places where the
reference tree is not on disk (GPU box, bench, smoke).  The reference's own config files load
unchanged through `Config.fromfile` (tests/test_configs_cpu.py does that when /root/reference is
present); these are the same numbers restated: configs/pillarnet/pillarnet_centerhead_nusc.py:6-81
and configs/pillarnet/pillarnet34_fpn_centerhead_waymo.py."""
import itertools
import logging

from .config import ConfigDict

NUSC_TASKS = [
    ["car"], ["truck", "construction_vehicle"], ["bus", "trailer"], ["barrier"],
    ["motorcycle", "bicycle"], ["pedestrian", "traffic_cone"],
]


def nusc_pillarnet18(double_flip=False):
    tasks = [dict(stride=8, class_names=names) for names in NUSC_TASKS]
    pillar, rng = 0.075, [-54, -54, -5.0, 54, 54, 3.0]
    model = dict(
        type="PillarNet", pretrained=None,
        reader=dict(type="DynamicPFE", in_channels=5, num_filters=(32,), pillar_size=pillar, pc_range=rng),
        backbone=dict(type="PillarResNet18", in_channels=32),
        neck=dict(type="RPNV1", layer_nums=[5, 5], num_filters=256, in_channels=[256, 256],
                  logger=logging.getLogger("RPN")),
        bbox_head=dict(type="CenterHead", tasks=tasks, in_channels=[256],
                       code_weights=[1.0] * 6 + [0.2, 0.2, 1.0, 1.0],
                       common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2),
                                     "vel": (2, 2)},
                       reg_iou="GIoU", pillar_size=pillar, point_cloud_range=rng))
    train_cfg = dict(assigner=dict(target_assigner=dict(tasks=tasks), dense_reg=1, gaussian_overlap=0.1,
                                   max_objs=500, min_radius=2, pc_range=rng, pillar_size=pillar),
                     hm_weight=1, bbox_weight=0.25, iou_weight=1, reg_iou_weight=0.25)
    test_cfg = dict(nms=dict(use_rotate_nms=True, nms_pre_max_size=1000, nms_post_max_size=83,
                             nms_iou_threshold=0.2),
                    rectifier=0, score_threshold=0.1, double_flip=double_flip,
                    post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0])
    return ConfigDict(model=model, train_cfg=train_cfg, test_cfg=test_cfg,
                      class_names=list(itertools.chain(*NUSC_TASKS)), pillar_size=pillar,
                      point_cloud_range=rng)


def waymo_pillarnet34_fpn():
    tasks = [dict(stride=8, class_names=["VEHICLE"]), dict(stride=4, class_names=["PEDESTRIAN", "CYCLIST"])]
    pillar, rng = 0.1, [-75.2, -75.2, -2, 75.2, 75.2, 4.0]
    model = dict(
        type="PillarNet",
        reader=dict(type="DynamicPFE", in_channels=5, num_filters=(32,), pillar_size=pillar, pc_range=rng),
        backbone=dict(type="PillarResNet34", in_channels=32),
        neck=dict(type="RPNG", layer_nums=[5, 5], num_filters=[256, 128], in_channels=[256, 256, 128],
                  logger=logging.getLogger("RPN")),
        bbox_head=dict(type="CenterHead", tasks=tasks, in_channels=[256, 128],
                       code_weights=[1.0] * 8,
                       common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2),
                                     "iou": (1, 2)},
                       reg_iou="GIoU", pillar_size=pillar, point_cloud_range=rng))
    test_cfg = dict(nms=dict(use_rotate_nms=False, use_multi_class_nms=True,
                             nms_pre_max_size=[2048, 1024, 1024], nms_post_max_size=[200, 150, 150],
                             nms_iou_threshold=[0.8, 0.55, 0.55]),
                    rectifier=[0.0, 0.0, 0.0], score_threshold=0.1,
                    post_center_limit_range=[-80, -80, -10.0, 80, 80, 10.0])
    return ConfigDict(model=model, train_cfg=dict(), test_cfg=test_cfg, pillar_size=pillar,
                      point_cloud_range=rng)


def waymo_pillarrcnn_fpn():
    """configs/pillarrcnn/pillarrcnn_fpn_centerhead_waymo.py:4-139 (BASELINE configs[4])."""
    first = waymo_pillarnet34_fpn()
    m = first.model
    m["backbone"] = dict(type="PillarResNet18", in_channels=32)
    m["bbox_head"]["common_heads"] = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2)}
    pillar, rng = first.pillar_size, first.point_cloud_range
    roi_cfg = dict(CLASS_AGNOSTIC=True, SHARED_FC=[256, 256], CLS_FC=[256, 256], REG_FC=[256, 256], DP_RATIO=0.3,
                   TARGET_CONFIG=dict(ROI_PER_IMAGE=128, FG_RATIO=0.5, SAMPLE_ROI_BY_EACH_CLASS=True,
                                      CLS_SCORE_TYPE="roi_iou", CLS_FG_THRESH=0.7, CLS_BG_THRESH=0.25,
                                      CLS_BG_THRESH_LO=0.1, HARD_BG_RATIO=0.8, REG_FG_THRESH=0.5),
                   LOSS_CONFIG=dict(CLS_LOSS="BinaryCrossEntropy", REG_LOSS="L1",
                                    LOSS_WEIGHTS={"rcnn_cls_weight": 1.0, "rcnn_reg_weight": 1.0,
                                                  "code_weights": [1.0] * 7}))
    model = dict(
        type="PillarRCNN", freeze=False, first_stage_cfg=m,
        second_stage_modules=[dict(type="BEVStrideFeature", feature_sources=["conv3"], grid_size=7, out_stride=4,
                                   in_channels=128, share_channels=64, pillar_size=pillar, pc_range=rng)],
        point_head=dict(type="PointHead", in_channels=64, num_class=1,
                        model_cfg=dict(CLASS_AGNOSTIC=True, CLS_FC=[256, 256],
                                       TARGET_CONFIG=dict(GT_EXTRA_WIDTH=[0.2, 0.2, 0.2]),
                                       LOSS_CONFIG=dict(LOSS_REG="smooth-l1",
                                                        LOSS_WEIGHTS={"point_cls_weight": 1.0}))),
        roi_head=dict(type="RoIMIXHead", in_channels=64, mixer_type="", num_patches=49, model_cfg=roi_cfg, code_size=7))
    test_cfg = ConfigDict(post_center_limit_range=[-80, -80, -10.0, 80, 80, 10.0],
                          nms=dict(use_multi_class_nms=True, nms_pre_max_size=[2048, 1024, 1024],
                                   nms_post_max_size=[200, 150, 150], nms_iou_threshold=[0.8, 0.55, 0.55]),
                          score_threshold=0.1, rectifier=[0.0, 0.0, 0.0])
    return ConfigDict(model=model, train_cfg=dict(), test_cfg=test_cfg, pillar_size=pillar, point_cloud_range=rng)


def build(preset):
    """preset ConfigDict -> detector (train/test cfg wired as tools/train.py:148 does)."""
    from .builder import build_detector
    return build_detector(preset.model, train_cfg=preset.train_cfg, test_cfg=preset.test_cfg)
