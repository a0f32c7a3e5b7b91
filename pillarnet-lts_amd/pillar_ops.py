"""Host side of the point -> pillar operators (mirror of det3d/ops/pillar_ops).

`PillarMaxPooling` keeps the reference module's constructor, sub-module names and state-dict keys
(`shared_mlps.0.weight [C, F+2]`, `shared_mlps.1.{weight,bias,running_mean,running_var}`,
pillar_modules.py:8-54); its compute is one call into the HIP library (`pn_pillarize`).
"""
import ctypes
from typing import List

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .spconv import SparseConvTensor


def bev_spatial_shape(pillar_size, point_cloud_range):
    """pillar_utils.py:7-10."""
    W = round((point_cloud_range[3] - point_cloud_range[0]) / pillar_size)
    H = round((point_cloud_range[4] - point_cloud_range[1]) / pillar_size)
    return int(H), int(W)


def make_grid(batch, pillar_size, point_cloud_range, point_feats, out_channels, true_divide=False):
    H, W = bev_spatial_shape(pillar_size, point_cloud_range)
    g = _lib.pnPillarGrid()
    g.batch, g.height, g.width = batch, H, W
    g.point_feats, g.out_channels = point_feats, out_channels
    g.x_min = float(np.float32(point_cloud_range[0]))
    g.y_min = float(np.float32(point_cloud_range[1]))
    g.inv_pillar = float(np.float32(1.0) / np.float32(pillar_size))
    g.pillar_size = float(np.float32(pillar_size))
    # pillar_utils.py:19-20: python-double sum, rounded once when torch makes it an fp32 scalar
    g.x_offset = float(np.float32(pillar_size / 2.0 + point_cloud_range[0]))
    g.y_offset = float(np.float32(pillar_size / 2.0 + point_cloud_range[1]))
    g.true_divide = 1 if true_divide else 0
    return g


def fold_batchnorm(bn: nn.modules.batchnorm._BatchNorm, conv_bias=None):
    """eval-mode BN (+ optional preceding bias) as per-channel fp32 scale/shift."""
    with torch.no_grad():
        var = bn.running_var.double()
        mean = bn.running_mean.double()
        gamma = bn.weight.double() if bn.weight is not None else torch.ones_like(var)
        beta = bn.bias.double() if bn.bias is not None else torch.zeros_like(var)
        scale = gamma / torch.sqrt(var + bn.eps)
        if conv_bias is not None:
            mean = mean - conv_bias.double()
        shift = beta - mean * scale
    return scale.float().contiguous(), shift.float().contiguous()


def pillarize(points: List[torch.Tensor], weight, scale, shift, pillar_size, point_cloud_range,
              want_point_index=False, true_divide=False):
    """Fused a1-a4 on the device.  points: list of [Ni, F] fp32 CUDA tensors (one per sample).

    Returns (SparseConvTensor with fp32 [cap, C] features, point_pillar_index or None).
    """
    assert len(points) >= 1
    dev = points[0].device
    if dev.type != "cuda":
        raise _lib.PillarnetHipError("pillarize: points must live on the GPU (no CPU fallback)")
    B = len(points)
    F = points[0].shape[1]
    C = weight.shape[0]
    assert weight.shape[1] == F + 2, (weight.shape, F)
    pts = points[0] if B == 1 else torch.cat(points, dim=0)
    pts = pts.contiguous().float()
    n = pts.shape[0]
    offs = np.zeros(B + 1, dtype=np.int32)
    offs[1:] = np.cumsum([p.shape[0] for p in points])
    g = make_grid(B, pillar_size, point_cloud_range, F, C, true_divide)
    L = _lib.lib()
    ws_bytes = L.pn_pillarize_workspace_bytes(ctypes.byref(g), n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    cap = max(1, min(n, B * g.height * g.width))
    indices = torch.empty((cap, 3), dtype=torch.int32, device=dev)
    feats = torch.empty((cap, C), dtype=torch.float32, device=dev)
    num = torch.empty(1, dtype=torch.int32, device=dev)   # always written by the emit kernel
    ppi = torch.empty(n, dtype=torch.int32, device=dev) if want_point_index else None
    bits_p, rank_p = ctypes.c_void_p(), ctypes.c_void_p()
    rc = L.pn_pillarize(ctypes.byref(g), _lib.ptr(pts), n,
                        offs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                        _lib.ptr(weight), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(indices),
                        _lib.ptr(feats), _lib.ptr(ppi), _lib.ptr(num), _lib.ptr(ws), ws_bytes,
                        ctypes.byref(bits_p), ctypes.byref(rank_p), _lib.current_stream(dev))
    _lib.check(rc, "pn_pillarize")
    words = (B * g.height * g.width + 63) // 64
    base = ws.data_ptr()
    occ_bits = ws[bits_p.value - base: bits_p.value - base + words * 8].view(torch.int64)
    occ_rank = ws[rank_p.value - base: rank_p.value - base + words * 4].view(torch.int32)
    sp = SparseConvTensor(feats, indices, (g.height, g.width), B, num, occ_bits, occ_rank,
                          keepalive=(ws, pts))
    return sp, ppi


class PillarQueryAndGroup(nn.Module):
    """pillar_utils.py:13-20: carries the grid geometry (no parameters)."""

    def __init__(self, pillar_size, point_cloud_range):
        super().__init__()
        self.pillar_size = pillar_size
        self.point_cloud_range = list(point_cloud_range)
        self.spatial_shape = bev_spatial_shape(pillar_size, point_cloud_range)
        self.x_offset = pillar_size / 2. + point_cloud_range[0]
        self.y_offset = pillar_size / 2. + point_cloud_range[1]


class PillarMaxPooling(nn.Module):
    """pillar_modules.py:8-74.  forward() takes the raw per-sample point tensors: the cell
    coordinates, range mask and batch concat of DynamicPFE.forward are part of the fused kernel."""

    def __init__(self, mlps: List[int], pillar_size: float, point_cloud_range: List[float],
                 activation='relu'):
        super().__init__()
        self.groups = PillarQueryAndGroup(pillar_size, point_cloud_range)
        self.height, self.width = self.groups.spatial_shape
        if activation == 'relu':
            act = nn.ReLU
        elif activation == 'silu':
            act = nn.SiLU
        else:
            raise NotImplementedError
        self.activation = activation
        shared_mlp = []
        for k in range(len(mlps) - 1):
            shared_mlp.extend([nn.Linear(mlps[k], mlps[k + 1], bias=False),
                               nn.BatchNorm1d(mlps[k + 1], momentum=0.01, eps=1e-3), act()])
        self.shared_mlps = nn.Sequential(*shared_mlp)
        self.init_weights(weight_init='kaiming')
        self._folded = None

    def init_weights(self, weight_init='xavier'):
        # pillar_modules.py:35-54
        if weight_init == 'kaiming':
            init_func = nn.init.kaiming_normal_
        elif weight_init == 'xavier':
            init_func = nn.init.xavier_normal_
        elif weight_init == 'normal':
            init_func = nn.init.normal_
        else:
            raise NotImplementedError
        for m in self.modules():
            if isinstance(m, nn.Linear):
                if weight_init == 'normal':
                    init_func(m.weight, mean=0, std=0.001)
                else:
                    init_func(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _fold(self):
        lin, bn = self.shared_mlps[0], self.shared_mlps[1]
        key = (lin.weight._version, bn.weight._version, bn.bias._version,
               bn.running_mean._version, bn.running_var._version, lin.weight.device)
        if self._folded is None or self._folded[0] != key:
            scale, shift = fold_batchnorm(bn)
            self._folded = (key, lin.weight.detach().float().contiguous(), scale, shift)
        return self._folded[1:]

    def forward(self, points: List[torch.Tensor], want_point_index=False):
        if len(self.shared_mlps) != 3 or self.activation != 'relu':
            raise NotImplementedError("the fused HIP reader implements one Linear+BN1d+ReLU layer")
        if self.training:
            raise NotImplementedError("the fused reader is inference-only; training goes through forward_unfused")
        w, scale, shift = self._fold()
        sp, ppi = pillarize(points, w, scale, shift, self.groups.pillar_size,
                            self.groups.point_cloud_range, want_point_index)
        return (sp, ppi) if want_point_index else sp


# =================================================================================================
# Unfused / training path: the reference's operator surface (det3d/ops/pillar_ops/{group_utils,
# scatter_utils,pillar_utils}.py) on the HIP C ABI, differentiable where the reference's is.
# =================================================================================================
class pillar_cuda:
    """Drop-in for the reference's `pillar_cuda` pybind module (pillar_api.cpp:10-21): same function
    names, argument order and out-parameter style; every call returns 1."""

    @staticmethod
    def create_point_pillar_index_stack_wrapper(pts_xy, pts_batch_cnt, pillars_mask, point_pillar_index):
        B, H, W = pillars_mask.shape
        rc = _lib.lib().pn_create_point_pillar_index_stack(
            _lib.ptr(pts_xy), _lib.ptr(pts_batch_cnt), pts_xy.shape[0], B, H, W, _lib.ptr(pillars_mask),
            _lib.ptr(point_pillar_index), _lib.current_stream(pts_xy.device))
        _lib.check(rc, "pn_create_point_pillar_index_stack")
        return 1

    @staticmethod
    def create_pillar_indices_wrapper(pillars_position, pillar_indices):
        B, H, W = pillars_position.shape
        rc = _lib.lib().pn_create_pillar_indices(_lib.ptr(pillars_position), B, H, W, _lib.ptr(pillar_indices),
                                                 _lib.current_stream(pillars_position.device))
        _lib.check(rc, "pn_create_pillar_indices")
        return 1

    @staticmethod
    def gather_indice_wrapper(index, indices, outs):
        rc = _lib.lib().pn_gather_indice(_lib.ptr(index), index.shape[0], _lib.ptr(indices), _lib.ptr(outs),
                                         _lib.current_stream(index.device))
        _lib.check(rc, "pn_gather_indice")
        return 1

    @staticmethod
    def gather_feature_wrapper(index, features, outs):
        rc = _lib.lib().pn_gather_feature(_lib.ptr(index), index.shape[0], _lib.ptr(features), features.shape[1],
                                          _lib.ptr(outs), _lib.current_stream(index.device))
        _lib.check(rc, "pn_gather_feature")
        return 1

    @staticmethod
    def gather_feature_grad_wrapper(grad_outs, index, grad_features):
        rc = _lib.lib().pn_gather_feature_grad(_lib.ptr(grad_outs), _lib.ptr(index), index.shape[0], grad_outs.shape[1],
                                               _lib.ptr(grad_features), _lib.current_stream(grad_outs.device))
        _lib.check(rc, "pn_gather_feature_grad")
        return 1

    @staticmethod
    def scatter_max_wrapper(index, src, arg, out):
        L, C = src.shape
        M = out.shape[0]
        Lb = _lib.lib()
        wsb = Lb.pn_scatter_max_workspace_bytes(max(M, 1), C)
        ws = torch.empty(wsb, dtype=torch.uint8, device=src.device)
        rc = Lb.pn_scatter_max(_lib.ptr(index), _lib.ptr(src), L, C, M, _lib.ptr(arg), _lib.ptr(out), _lib.ptr(ws),
                               wsb, _lib.current_stream(src.device))
        _lib.check(rc, "pn_scatter_max")
        return 1

    @staticmethod
    def scatter_max_grad_wrapper(grad_out, arg, grad_src):
        M, C = grad_out.shape
        rc = _lib.lib().pn_scatter_max_grad(_lib.ptr(grad_out), _lib.ptr(arg), M, C, _lib.ptr(grad_src),
                                            _lib.current_stream(grad_out.device))
        _lib.check(rc, "pn_scatter_max_grad")
        return 1


def _cuda_i32(t, what):
    if t.device.type != "cuda":
        raise _lib.PillarnetHipError(f"{what}: tensor must live on the GPU (no CPU fallback)")
    return t.contiguous()


class ScatterMaxFunction(torch.autograd.Function):
    """scatter_utils.py:7-35.  `arg` ties resolve to the lowest point index (deterministic)."""

    @staticmethod
    def forward(ctx, src, index, M):
        src, index = _cuda_i32(src, "scatter_max").float(), _cuda_i32(index, "scatter_max").int()
        L, C = src.shape
        arg = index.new_full((M, C), -1)
        out = src.new_zeros(M, C)
        pillar_cuda.scatter_max_wrapper(index, src, arg, out)
        ctx.for_backwards = (L, C, arg)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        L, C, arg = ctx.for_backwards
        grad_src = grad_out.new_zeros(L, C)
        pillar_cuda.scatter_max_grad_wrapper(grad_out.contiguous(), arg, grad_src)
        return grad_src, None, None


scatter_max = ScatterMaxFunction.apply


def gather_indice(indices, index):
    """group_utils.py:6-13: outs[i] = indices[index[i]]."""
    outs = indices.new_zeros((index.shape[0],))
    pillar_cuda.gather_indice_wrapper(_cuda_i32(index, "gather_indice"), _cuda_i32(indices, "gather_indice"), outs)
    return outs


class GatherFeature(torch.autograd.Function):
    """group_utils.py:16-36: outs[l] = features[index[l]]; the backward adds grad_outs[l] into row index[l]
    (the reference's own backward passes its arguments in the wrong order and is unreachable: its one caller
    detaches the input, pillar_modules.py:71)."""

    @staticmethod
    def forward(ctx, features, index):
        index = _cuda_i32(index, "gather_feature")
        feats = _cuda_i32(features.detach(), "gather_feature")
        outs = feats.new_zeros((index.shape[0], feats.shape[1]))
        pillar_cuda.gather_feature_wrapper(index, feats, outs)
        ctx.for_backwards = (feats.shape[0], feats.shape[1], index)
        return outs

    @staticmethod
    def backward(ctx, grad_outs):
        N, C, index = ctx.for_backwards
        grad_features = grad_outs.new_zeros(N, C)
        pillar_cuda.gather_feature_grad_wrapper(grad_outs.contiguous().float(), index, grad_features)
        return grad_features, None


def gather_feature(features, index):
    return GatherFeature.apply(features, index)


def query_and_group(groups: PillarQueryAndGroup, pts_xy, pts_batch_cnt, pts_features):
    """PillarQueryAndGroup.forward (pillar_utils.py:22-58), unfused, with the reference's own flow:
    bool mask -> cumsum -> position map -> (b,y,x) rows -> per-point pillar id -> centre offsets."""
    batch = pts_batch_cnt.numel()
    H, W = groups.spatial_shape
    n = pts_xy.shape[0]
    ppi_flat = pts_batch_cnt.new_full((n,), -1)
    mask = torch.zeros((batch, H, W), dtype=torch.bool, device=pts_xy.device)
    pillar_cuda.create_point_pillar_index_stack_wrapper(pts_xy.contiguous(), pts_batch_cnt, mask, ppi_flat)
    pos = torch.cumsum(mask.view(-1), dim=0, dtype=torch.int32)
    out_dim = int(pos[-1].item())
    pos = (pos.view(batch, H, W) * mask - 1).contiguous()
    pillar_indices = pts_batch_cnt.new_zeros((out_dim, 3))
    pillar_cuda.create_pillar_indices_wrapper(pos, pillar_indices)
    point_pillar_indices = gather_indice(pos.view(-1), ppi_flat)
    centers = torch.stack((pillar_indices[:, 2] * groups.pillar_size + groups.x_offset,
                           pillar_indices[:, 1] * groups.pillar_size + groups.y_offset), dim=1)
    ctr = gather_feature(centers, point_pillar_indices)
    feats = torch.cat([pts_features[:, :2] - ctr, pts_features], dim=1)
    return pillar_indices, point_pillar_indices, feats


def _pmp_forward_unfused(self, pts_xy, pts_batch_cnt, pts_feature):
    """PillarMaxPooling.forward with the reference's signature (pillar_modules.py:56-74): differentiable
    w.r.t. the MLP parameters, BatchNorm in whatever mode the module is in."""
    batch = pts_batch_cnt.shape[0]
    pillar_indices, ppi, feats = query_and_group(self.groups, pts_xy, pts_batch_cnt, pts_feature)
    y = self.shared_mlps(feats.detach())
    pooled = scatter_max(y, ppi, pillar_indices.shape[0])
    return SparseConvTensor(pooled, pillar_indices, (self.height, self.width), batch)


PillarMaxPooling.forward_unfused = _pmp_forward_unfused
