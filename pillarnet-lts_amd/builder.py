"""det3d/models/builder.py:17-54."""
from torch import nn

from .registry import (BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, POINT_HEAD, READERS, ROI_HEAD,
                       SECOND_STAGE, build_from_cfg)


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_reader(cfg):
    return build(cfg, READERS)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_second_stage_module(cfg):
    return build(cfg, SECOND_STAGE)


def build_point_head(cfg):
    return build(cfg, POINT_HEAD)


def build_roi_head(cfg):
    return build(cfg, ROI_HEAD)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
