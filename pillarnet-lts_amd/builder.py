"""Factory entry points with the reference's names (det3d/models/builder.py:17-54): one table from builder
name to registry; a list of configs becomes an `nn.Sequential` of the built modules."""
import functools

from torch import nn

from . import registry as _reg


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[_reg.build_from_cfg(c, registry, default_args) for c in cfg])
    return _reg.build_from_cfg(cfg, registry, default_args)


_TABLE = dict(reader=_reg.READERS, backbone=_reg.BACKBONES, neck=_reg.NECKS, head=_reg.HEADS, loss=_reg.LOSSES,
              second_stage_module=_reg.SECOND_STAGE, point_head=_reg.POINT_HEAD, roi_head=_reg.ROI_HEAD)
for _name, _registry in _TABLE.items():
    globals()["build_" + _name] = functools.partial(lambda cfg, _r: build(cfg, _r), _r=_registry)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, _reg.DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
