"""NECKS registry: RPN, RPNV1, RPNV2, RPNG, RPNGV2 (det3d/models/necks/rpn.py:14-446).  All dense
nn.Conv2d / ConvTranspose2d / BatchNorm2d(eps 1e-3, momentum 0.01) / ReLU stacks -- they stay on
PyTorch-ROCm (MIOpen).  Sub-module names follow the reference so checkpoints load."""
import logging

import numpy as np
import torch
import torch.nn as nn

from .nn_utils import Sequential, build_norm_layer, xavier_init
from .registry import NECKS
from .spconv import SparseConvTensor

_BN = dict(type="BN", eps=1e-3, momentum=0.01)


def _stack(norm_cfg, inplanes, planes, num_blocks, stride=1):
    """ZeroPad2d(1) + 3x3 conv (no padding) + BN + ReLU, then num_blocks x (3x3 conv, BN, ReLU)
    -- rpn.py:104-117 / 172-185."""
    block = Sequential(nn.ZeroPad2d(1), nn.Conv2d(inplanes, planes, 3, stride=stride, bias=False),
                       build_norm_layer(norm_cfg, planes)[1], nn.ReLU())
    for _ in range(num_blocks):
        block.add(nn.Conv2d(planes, planes, 3, padding=1, bias=False))
        block.add(build_norm_layer(norm_cfg, planes)[1])
        block.add(nn.ReLU())
    return block


def _up2(norm_cfg, cin, cout):
    return Sequential(nn.ConvTranspose2d(cin, cout, 2, stride=2, bias=False),
                      build_norm_layer(norm_cfg, cout)[1], nn.ReLU())


def _reduce(norm_cfg, cin, cout):
    return Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False),
                      build_norm_layer(norm_cfg, cout)[1], nn.ReLU())


def _dense(x, like=None):
    if isinstance(x, SparseConvTensor):
        x = x.dense()
    if like is not None and x.dtype != like.dtype:
        x = x.to(like.dtype)
    return x


class _NeckBase(nn.Module):
    @property
    def downsample_factor(self):
        return 1

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution="uniform")

    def _levels(self, feats, names):
        ref = None if getattr(self, "_dense_hip", False) else next(self.parameters())
        return [_dense(feats[n], ref) for n in names]

    # ---- HIP dense path (bf16, eval): every conv is one gather-GEMM launch, concat is free --------
    def _use_hip(self, x):
        return getattr(self, "_dense_hip", False) and not self.training and x.dtype == torch.bfloat16

    def _plan(self, name):
        from . import dense_hip
        return dense_hip.PLANS.get(getattr(self, name), dense_hip.compile_stack)

    @staticmethod
    def _concat_target(lateral, extra_channels):
        """Rows view of [lateral | room for extra_channels]: in place when `lateral` already sits in
        a wide enough buffer (SparseConvTensor.dense(out_channels=...)), else one copy."""
        from . import dense_hip
        r = dense_hip.Rows.from_nchw(lateral)
        C = r.t.shape[1]
        if r.ld >= C + extra_channels:
            return r.widen(C + extra_channels), C
        buf = torch.empty((r.t.shape[0], C + extra_channels), dtype=r.t.dtype, device=r.t.device)
        buf[:, :C].copy_(r.t)
        return dense_hip.Rows(buf, r.B, r.H, r.W), C


@NECKS.register_module
class RPNV1(_NeckBase):
    """rpn.py:138-207: conv5 stack -> 2x up -> cat conv4 -> stack (nuScenes, stride-8 head)."""

    def __new__(cls, layer_nums=None, num_filters=None, in_channels=None, norm_cfg=None, logger=None, **kwargs):
        if "ds_num_filters" in kwargs:       # legacy kwargs (configs/nusc/**): legacy.legacy_rpn_v translates
            from .legacy import legacy_rpn_v
            return legacy_rpn_v("RPNV1", layer_nums, dict(kwargs, logger=logger))
        return super().__new__(cls)

    def __init__(self, layer_nums, num_filters=None, in_channels=None, norm_cfg=None, logger=None, **kwargs):
        if "ds_num_filters" in kwargs:
            return                           # built by __new__ from the translated kwargs
        super().__init__()
        self.norm_cfg = norm_cfg or dict(type="BN", momentum=0.01, eps=1e-3)
        self.block_5 = _stack(self.norm_cfg, in_channels[0], in_channels[0], layer_nums[0])
        self.deblock_5 = _up2(self.norm_cfg, in_channels[0], in_channels[1])
        self.block_4 = _stack(self.norm_cfg, in_channels[1] * 2, num_filters, layer_nums[1])
        (logger or logging.getLogger("RPN")).info("Finish RPN Initialization")

    def forward(self, pillar_features, **kwargs):
        c4, c5 = self._levels(pillar_features, ("conv4", "conv5"))
        if self._use_hip(c4):
            from . import dense_hip
            x5 = dense_hip.run_stack(self._plan("block_5"), dense_hip.Rows.from_nchw(c5))
            up = self._plan("deblock_5")
            cat, C = self._concat_target(c4, up[-1].cout)
            dense_hip.run_stack(up, x5, last_out=cat.t[:, C:])
            return (dense_hip.run_stack(self._plan("block_4"), cat).nchw(),)
        x = torch.cat([c4, self.deblock_5(self.block_5(c5))], dim=1)
        return (self.block_4(x),)


@NECKS.register_module
class RPNV2(_NeckBase):
    """rpn.py:211-272: same shape one level down (conv4 stack -> up -> cat conv3)."""

    def __new__(cls, layer_nums=None, in_channels=None, num_filters=None, norm_cfg=None, logger=None, **kwargs):
        if "ds_num_filters" in kwargs:       # legacy kwargs: legacy.legacy_rpn_v translates
            from .legacy import legacy_rpn_v
            return legacy_rpn_v("RPNV2", layer_nums, dict(kwargs, logger=logger))
        return super().__new__(cls)

    def __init__(self, layer_nums, in_channels=None, num_filters=None, norm_cfg=None, logger=None, **kwargs):
        if "ds_num_filters" in kwargs:
            return                           # built by __new__ from the translated kwargs
        super().__init__()
        self.norm_cfg = norm_cfg or dict(type="BN", momentum=0.01, eps=1e-3)
        self.block_4 = _stack(self.norm_cfg, in_channels[0], in_channels[0], layer_nums[0])
        self.deblock_4 = _up2(self.norm_cfg, in_channels[0], in_channels[1])
        self.block_3 = _stack(self.norm_cfg, in_channels[1] * 2, num_filters, layer_nums[1])
        (logger or logging.getLogger("RPN")).info("Finish RPN Initialization")

    def forward(self, pillar_features, **kwargs):
        c3, c4 = self._levels(pillar_features, ("conv3", "conv4"))
        x = torch.cat([c3, self.deblock_4(self.block_4(c4))], dim=1)
        return (self.block_3(x),)


@NECKS.register_module
class RPNG(_NeckBase):
    """rpn.py:276-355: two-scale top-down FPN; returns (stride-8 map, stride-4 map)."""

    def __init__(self, layer_nums, in_channels, num_filters, norm_cfg=None, logger=None):
        super().__init__()
        self.norm_cfg = norm_cfg or _BN
        self.block_5 = _stack(self.norm_cfg, in_channels[0], in_channels[0], layer_nums[0])
        self.top_down_54 = _up2(self.norm_cfg, in_channels[0], in_channels[1])
        self.block_4 = _stack(self.norm_cfg, in_channels[1] * 2, num_filters[0], layer_nums[0])
        self.top_down_43 = _up2(self.norm_cfg, num_filters[0], in_channels[2])
        self.block_3 = _stack(self.norm_cfg, in_channels[2] * 2, num_filters[1], layer_nums[1])
        (logger or logging.getLogger("RPN")).info("Finish RPN Initialization")

    def forward(self, pillar_features, **kwargs):
        c3, c4, c5 = self._levels(pillar_features, ("conv3", "conv4", "conv5"))
        if self._use_hip(c4):
            from . import dense_hip
            x5 = dense_hip.run_stack(self._plan("block_5"), dense_hip.Rows.from_nchw(c5))
            up = self._plan("top_down_54")
            cat4, C4 = self._concat_target(c4, up[-1].cout)
            dense_hip.run_stack(up, x5, last_out=cat4.t[:, C4:])
            x4 = dense_hip.run_stack(self._plan("block_4"), cat4)
            up = self._plan("top_down_43")
            cat3, C3 = self._concat_target(c3, up[-1].cout)
            dense_hip.run_stack(up, x4, last_out=cat3.t[:, C3:])
            x3 = dense_hip.run_stack(self._plan("block_3"), cat3)
            return (x4.nchw(), x3.nchw())
        x4 = self.block_4(torch.cat([c4, self.top_down_54(self.block_5(c5))], dim=1))
        x3 = self.block_3(torch.cat([c3, self.top_down_43(x4)], dim=1))
        return (x4, x3)


@NECKS.register_module
class RPNGV2(_NeckBase):
    """rpn.py:358-446: RPNG with 3x3 lateral `reduce_*` convs and half-width top-down paths."""

    def __init__(self, layer_nums, in_channels, num_filters, norm_cfg=None, logger=None):
        super().__init__()
        self.norm_cfg = norm_cfg or _BN
        self.block_5 = _stack(self.norm_cfg, in_channels[0], in_channels[0], layer_nums[0])
        self.top_down_54 = _up2(self.norm_cfg, in_channels[0], num_filters[0] // 2)
        self.reduce_4 = _reduce(self.norm_cfg, in_channels[1], num_filters[0] // 2)
        self.block_4 = _stack(self.norm_cfg, num_filters[0], num_filters[0], layer_nums[0])
        self.top_down_43 = _up2(self.norm_cfg, num_filters[0], num_filters[1] // 2)
        self.reduce_3 = _reduce(self.norm_cfg, in_channels[2], num_filters[1] // 2)
        self.block_3 = _stack(self.norm_cfg, num_filters[1], num_filters[1], layer_nums[1])
        (logger or logging.getLogger("RPN")).info("Finish RPN Initialization")

    def forward(self, pillar_features, **kwargs):
        c3, c4, c5 = self._levels(pillar_features, ("conv3", "conv4", "conv5"))
        x4 = self.block_4(torch.cat([self.reduce_4(c4), self.top_down_54(self.block_5(c5))], dim=1))
        x3 = self.block_3(torch.cat([self.reduce_3(c3), self.top_down_43(x4)], dim=1))
        return (x4, x3)


@NECKS.register_module
class RPN(nn.Module):
    """rpn.py:14-134: the SECOND-style multi-stride RPN (legacy configs); takes a dense tensor."""

    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters,
                 in_channels=None, norm_cfg=None, logger=None, **kwargs):
        super().__init__()
        if in_channels is None:
            in_channels = kwargs.pop("num_input_features")      # the legacy configs' name for it
        self._layer_strides, self._num_filters, self._layer_nums = ds_layer_strides, ds_num_filters, layer_nums
        self._upsample_strides, self._num_upsample_filters = us_layer_strides, us_num_filters
        self._norm_cfg = norm_cfg or _BN
        assert len(ds_layer_strides) == len(layer_nums) == len(ds_num_filters)
        assert len(us_num_filters) == len(us_layer_strides)
        self._upsample_start_idx = len(layer_nums) - len(us_layer_strides)
        ratios = [us_layer_strides[i] / np.prod(ds_layer_strides[: i + self._upsample_start_idx + 1])
                  for i in range(len(us_layer_strides))]
        assert all(r == ratios[0] for r in ratios)
        in_filters = [in_channels, *ds_num_filters[:-1]]
        blocks, deblocks = [], []
        for i, n in enumerate(layer_nums):
            blocks.append(_stack(self._norm_cfg, in_filters[i], ds_num_filters[i], n, ds_layer_strides[i]))
            j = i - self._upsample_start_idx
            if j >= 0:
                s = us_layer_strides[j]
                if s > 1:
                    up = nn.ConvTranspose2d(ds_num_filters[i], us_num_filters[j], s, stride=s, bias=False)
                else:
                    s = int(np.round(1 / s))
                    up = nn.Conv2d(ds_num_filters[i], us_num_filters[j], s, stride=s, bias=False)
                deblocks.append(Sequential(up, build_norm_layer(self._norm_cfg, us_num_filters[j])[1],
                                           nn.ReLU()))
        self.blocks, self.deblocks = nn.ModuleList(blocks), nn.ModuleList(deblocks)
        (logger or logging.getLogger("RPN")).info("Finish RPN Initialization")

    @property
    def downsample_factor(self):
        f = np.prod(self._layer_strides)
        return f / self._upsample_strides[-1] if len(self._upsample_strides) > 0 else f

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution="uniform")

    def forward(self, x):
        if isinstance(x, dict):      # PillarNet hands over the backbone's stage dict: the stride-8 map
            x = x["conv4"]
            x = x.dense() if hasattr(x, "dense") else x
        ups = []
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if i - self._upsample_start_idx >= 0:
                ups.append(self.deblocks[i - self._upsample_start_idx](x))
        if ups:
            x = torch.cat(ups, dim=1)
        return tuple(x)  # sic: rpn.py:134 splits the batch dimension
