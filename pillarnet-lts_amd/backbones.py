"""BACKBONES registry: PillarResNet18 / 18S / 34 / 34S (det3d/models/backbones/PillarResNet.py:
9,73,153,225) and their residual blocks (det3d/models/backbones/base.py:38-63,98-213).

Module tree and parameter names are the reference's (`conv1.0.conv0.0.weight`, `conv2.0.weight`,
`conv2.1.running_mean`, `conv5.3.0.weight` ...), so its checkpoints load key for key.  In eval
mode every `conv -> BatchNorm1d [-> ReLU]` and every block tail `conv -> BN -> + identity -> ReLU`
is one launch of the HIP implicit-GEMM kernel with the norm folded into the epilogue.
"""
import torch
import torch.nn as nn

from . import sparse_conv
from .nn_utils import build_norm_layer
from .registry import BACKBONES
from .sparse_conv import (SparseConv2d, SparseModule, SparseReLU, SparseSequential, SubMConv2d,
                          conv_forward)

_BN1D = dict(type="BN1d", momentum=0.01, eps=1e-3)
_BN2D = dict(type="BN", momentum=0.01, eps=1e-3)


def conv2D3x3(in_planes, out_planes, stride=1, dilation=1, indice_key=None, bias=True):
    """base.py:38-63: stride 1 -> submanifold conv, otherwise a regular (site-creating) sparse conv."""
    assert stride >= 1
    cls = SubMConv2d if stride == 1 else SparseConv2d
    return cls(in_planes, out_planes, kernel_size=3, stride=stride, dilation=dilation,
               padding=dilation, bias=bias, indice_key=indice_key)


def _conv_bn(planes, norm_cfg, indice_key, relu):
    layers = [conv2D3x3(planes, planes, indice_key=indice_key, bias=True),
              build_norm_layer(norm_cfg, planes)[1]]
    if relu:
        layers.append(SparseReLU())
    return SparseSequential(*layers)


class Sparse2DBasicBlock(SparseModule):
    """base.py:181-213: x -> conv1(conv,bn,relu) -> conv2(conv,bn) -> +x -> relu.
    (`bias = norm_cfg is not None` is always true in the reference, base.py:192.)"""
    expansion = 1

    def __init__(self, planes, norm_cfg=None, indice_key=None):
        super().__init__()
        norm_cfg = norm_cfg or _BN1D
        self.conv1 = _conv_bn(planes, norm_cfg, indice_key, relu=True)
        self.conv2 = _conv_bn(planes, norm_cfg, indice_key, relu=False)
        self.relu = nn.ReLU()

    def _tail(self, x, identity):
        out = self.conv1(x)
        return conv_forward(out, self.conv2[0], self.conv2[1], relu=True, residual=identity)

    def forward(self, x):
        return self._tail(x, x)


class Sparse2DBasicBlockV(Sparse2DBasicBlock):
    """base.py:145-178: an extra conv0(conv,bn) first; its output is the residual branch."""

    def __init__(self, planes, norm_cfg=None, indice_key=None):
        nn.Module.__init__(self)
        norm_cfg = norm_cfg or _BN1D
        self.conv0 = _conv_bn(planes, norm_cfg, indice_key, relu=False)
        self.conv1 = _conv_bn(planes, norm_cfg, indice_key, relu=True)
        self.conv2 = _conv_bn(planes, norm_cfg, indice_key, relu=False)
        self.relu = nn.ReLU()

    def forward(self, x):
        x = self.conv0(x)
        return self._tail(x, x)


def post_act_block_dense(in_channels, kernel_size, stride=1, padding=0, dilation=1, norm_cfg=None):
    """base.py:98-107 (the reference closes it with spconv's SparseReLU, which is ReLU on tensors)."""
    return nn.Sequential(
        nn.Conv2d(in_channels, in_channels, kernel_size, stride, padding=padding, dilation=dilation,
                  bias=False),
        build_norm_layer(norm_cfg, in_channels)[1],
        SparseReLU())


class _PillarResNet(nn.Module):
    DEPTHS = (2, 2, 2, 2)   # residual blocks per sparse stage (the first one of stage 1 is a BlockV)
    DENSE_TAIL = True       # conv4.dense() + dense conv5 at stride 16

    def __init__(self, in_channels=32, **kwargs):
        super().__init__()
        c = in_channels
        blocks = [Sparse2DBasicBlockV(c, norm_cfg=_BN1D, indice_key="res1")]
        blocks += [Sparse2DBasicBlock(c, norm_cfg=_BN1D, indice_key="res1")
                   for _ in range(self.DEPTHS[0] - 1)]
        self.conv1 = SparseSequential(*blocks)
        for stage in (2, 3, 4):
            cin, cout = c * 2 ** (stage - 2), c * 2 ** (stage - 1)
            layers = [SparseConv2d(cin, cout, 3, 2, padding=1, bias=False),
                      build_norm_layer(_BN1D, cout)[1], SparseReLU()]
            layers += [Sparse2DBasicBlock(cout, norm_cfg=_BN1D, indice_key=f"res{stage}")
                       for _ in range(self.DEPTHS[stage - 1])]
            setattr(self, f"conv{stage}", SparseSequential(*layers))
        self.backbone_channels = {"conv1": c, "conv2": 2 * c, "conv3": 4 * c, "conv4": 8 * c}
        self.backbone_strides = {"conv1": 1, "conv2": 2, "conv3": 4, "conv4": 8}
        if self.DENSE_TAIL:
            self.conv5 = nn.Sequential(
                nn.Conv2d(256, 256, 3, 2, padding=1, bias=False),
                build_norm_layer(_BN2D, 256)[1], nn.ReLU(),
                post_act_block_dense(256, 3, padding=1, norm_cfg=_BN2D),
                post_act_block_dense(256, 3, padding=1, norm_cfg=_BN2D))
            self.backbone_channels["conv5"] = 256
            self.backbone_strides["conv5"] = 16

    def forward(self, sp_tensor):
        feats = {}
        x = sp_tensor
        ahead = None
        if (sparse_conv.RULES_AHEAD and not self.training and x.features_cap.device.type == "cuda"
                and x.features_cap.dtype == torch.bfloat16):
            ahead = sparse_conv.rules_ahead(x, [(getattr(self, f"conv{s}")[0], f"res{s}") for s in (2, 3, 4)])
        for stage in (1, 2, 3, 4):
            if ahead is not None and stage > 1:
                x._ahead = ahead[stage - 2]
            x = getattr(self, f"conv{stage}")(x)
            feats[f"conv{stage}"] = x
        if self.DENSE_TAIL:
            if getattr(self, "_dense_hip", False) and x.features_cap.dtype == torch.bfloat16:
                from . import dense_hip
                # the rows land in the first C channels of a 2C-wide buffer: RPNV1 / RPNG put the
                # upsampled conv5 path right behind them, so the later concat needs no copy
                dense4 = x.dense(out_channels=2 * x.features_cap.shape[1])
                plan = dense_hip.PLANS.get(self.conv5, dense_hip.compile_stack)
                feats["conv4"] = dense4
                feats["conv5"] = dense_hip.run_stack(plan, dense_hip.Rows.from_nchw(dense4)).nchw()
                return feats
            dense4 = x.dense()
            p = next(self.conv5.parameters())
            if dense4.dtype != p.dtype:
                dense4 = dense4.to(p.dtype)
            feats["conv4"] = dense4
            feats["conv5"] = self.conv5(dense4)
        return feats


@BACKBONES.register_module
class PillarResNet18S(_PillarResNet):
    DEPTHS, DENSE_TAIL = (2, 2, 2, 2), False


@BACKBONES.register_module
class PillarResNet18(_PillarResNet):
    DEPTHS, DENSE_TAIL = (2, 2, 2, 2), True


@BACKBONES.register_module
class PillarResNet34S(_PillarResNet):
    DEPTHS, DENSE_TAIL = (3, 4, 6, 3), False


@BACKBONES.register_module
class PillarResNet34(_PillarResNet):
    DEPTHS, DENSE_TAIL = (3, 4, 6, 3), True
