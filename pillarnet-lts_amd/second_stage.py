"""Pillar R-CNN second stage (inference): SECOND_STAGE.BEVStrideFeature, ROI_HEAD.RoIMIXHead,
POINT_HEAD.PointHead, DETECTORS.PillarRCNN -- det3d/models/second_stage/bev_interpolation.py:162-308,
det3d/models/roi_heads/{roi_mix_head,roi_head_template,mlp_layers}.py,
det3d/models/point_heads/point_head_simple.py, det3d/models/detectors/pillar_rcnn.py.

Module trees / parameter names are the reference's.  The RoI pillar pooling (7x7 rotated grid +
bilinear interpolation) is one HIP launch (`pn_roi_grid_pool`); the small 1-D conv MLPs of the RoI
head stay torch ops.  Training: ProposalTargetLayer (HIP rotated 3-D IoU, the reference's RNG calls),
RoI class / box losses, the PointHead's grid-point targets (`pn_points_in_boxes_2d`) and focal loss, and
`combine_loss`; the pooled features are differentiable through a torch expression of the same bilinear
weights (`roi_grid_pool` itself is forward-only)."""
import numpy as np
import torch
import torch.nn as nn

import torch.nn.functional as Fn

from . import _lib, builder, dense_hip, iou3d_nms
from .detectors import BaseDetector
from .nn_utils import build_norm_layer
from .registry import DETECTORS, POINT_HEAD, ROI_HEAD, SECOND_STAGE
from .spconv import SparseConvTensor

_BN2 = dict(type="BN", momentum=0.01, eps=1e-3)
_BN1 = dict(type="BN1d", eps=1e-3, momentum=0.01)


def roi_grid_pool_rows(rois, rows, B, H, W, grid, cell_size, x_min, y_min, want_points=True):
    """rois [B,R,7+]; rows: the feature map as channels-last pixel rows [B*H*W, C] (fp32 or bf16, row stride
    >= C) -> ([B,R,G*G,C] fp32, [B,R,G*G,2] world xy of the grid points)."""
    if rois.device.type != "cuda" or rows.device.type != "cuda":
        raise _lib.PillarnetHipError("roi_grid_pool: tensors must live on the GPU (no CPU fallback)")
    assert rows.dim() == 2 and rows.stride(1) == 1 and rows.shape[0] == B * H * W and rois.shape[0] == B
    if rows.dtype not in (torch.bfloat16, torch.float32):
        rows = rows.float()
    dt = _lib.PN_DTYPE_BF16 if rows.dtype == torch.bfloat16 else _lib.PN_DTYPE_F32
    rois = rois.contiguous().float()
    R, C = rois.shape[1], rows.shape[1]
    out = torch.empty((B, R, grid * grid, C), dtype=torch.float32, device=rois.device)
    pts = torch.empty((B, R, grid * grid, 2), dtype=torch.float32, device=rois.device) if want_points else None
    rc = _lib.lib().pn_roi_grid_pool(dt, _lib.ptr(rois), rois.shape[2], B, R, grid, _lib.ptr(rows), rows.stride(0),
                                     H, W, C, float(np.float32(x_min)), float(np.float32(y_min)),
                                     float(np.float32(cell_size)), _lib.ptr(out), _lib.ptr(pts),
                                     _lib.current_stream(rois.device))
    _lib.check(rc, "pn_roi_grid_pool")
    return out, pts


def roi_grid_pool(rois, feature_map, grid, cell_size, x_min, y_min, want_points=True):
    """As roi_grid_pool_rows for a [B,C,H,W] map in any memory format."""
    if rois.device.type != "cuda":
        raise _lib.PillarnetHipError("roi_grid_pool: tensors must live on the GPU (no CPU fallback)")
    r = dense_hip.Rows.from_nchw(feature_map)
    return roi_grid_pool_rows(rois, r.t, r.B, r.H, r.W, grid, cell_size, x_min, y_min, want_points)


class RoIGridPoolFunction(torch.autograd.Function):
    """Differentiable RoI pillar pooling on the HIP kernels: forward = pn_roi_grid_pool on the fp32 pixel rows,
    backward = pn_roi_grid_pool_grad (the map's gradient; RoIs are first-stage boxes and carry none)."""

    @staticmethod
    def forward(ctx, rois, feature_map, grid, cell_size, x_min, y_min):
        B, C, H, W = feature_map.shape
        rows = feature_map.detach().permute(0, 2, 3, 1).reshape(B * H * W, C).float().contiguous()
        rois = rois.detach().contiguous().float()
        out, pts = roi_grid_pool_rows(rois, rows, B, H, W, grid, cell_size, x_min, y_min, True)
        ctx.save_for_backward(rois)
        ctx.geom = (B, C, H, W, grid, cell_size, x_min, y_min, feature_map.dtype)
        ctx.mark_non_differentiable(pts)
        return out, pts

    @staticmethod
    def backward(ctx, g, _g_pts):
        (rois,) = ctx.saved_tensors
        B, C, H, W, grid, cell_size, x_min, y_min, dtype = ctx.geom
        gm = torch.zeros((B, H * W, C), dtype=torch.float32, device=g.device)
        g = g.contiguous().float()
        rc = _lib.lib().pn_roi_grid_pool_grad(_lib.ptr(rois), rois.shape[2], B, rois.shape[1], grid, _lib.ptr(g), H, W, C,
                                              float(np.float32(x_min)), float(np.float32(y_min)),
                                              float(np.float32(cell_size)), _lib.ptr(gm), _lib.current_stream(g.device))
        _lib.check(rc, "pn_roi_grid_pool_grad")
        return None, gm.view(B, H, W, C).permute(0, 3, 1, 2).to(dtype), None, None, None, None


def roi_grid_pool_train(rois, feature_map, grid, cell_size, x_min, y_min):
    """training-mode pooling: ([B,R,G*G,C] fp32 with a gradient path to `feature_map`, grid points)."""
    if rois.requires_grad:
        raise _lib.PillarnetHipError("roi_grid_pool_train: RoIs are constants of the second stage (detach them)")
    return RoIGridPoolFunction.apply(rois, feature_map, grid, cell_size, x_min, y_min)


def roi_grid_pool_torch(rois, feature_map, grid, cell_size, x_min, y_min):
    """The same pooling as a differentiable torch expression (training: gradients reach the fused map through
    the gathers): center_to_grid_box2d + rotation_2d (box_torch_ops.py:220-251) and bilinear_interpolate_torch
    (center_utils.py:91-120), batched over samples."""
    B, R = rois.shape[:2]
    C, H, W = feature_map.shape[1:]
    dev = rois.device
    ii, jj = torch.meshgrid(torch.arange(grid, device=dev), torch.arange(grid, device=dev), indexing="ij")
    idx = torch.stack([ii.reshape(-1), jj.reshape(-1)], -1).float()                     # (i, j)
    flat = rois.reshape(B * R, -1).float()
    dims = flat[:, None, 3:5]
    local = (idx[None] + 0.5) / grid * dims - dims / 2
    c, s = torch.cos(flat[:, 6]), torch.sin(flat[:, 6])
    px = local[..., 0] * c[:, None] + local[..., 1] * s[:, None] + flat[:, None, 0]
    py = -local[..., 0] * s[:, None] + local[..., 1] * c[:, None] + flat[:, None, 1]
    pts = torch.stack([px, py], -1).view(B, R, grid * grid, 2)
    x = ((px - x_min) / cell_size).view(B, -1)
    y = ((py - y_min) / cell_size).view(B, -1)
    x0, y0 = torch.floor(x).long(), torch.floor(y).long()
    x1, y1 = x0 + 1, y0 + 1
    x0, x1 = x0.clamp(0, W - 1), x1.clamp(0, W - 1)
    y0, y1 = y0.clamp(0, H - 1), y1.clamp(0, H - 1)
    rows = feature_map.permute(0, 2, 3, 1).reshape(B, H * W, C).float()
    take = lambda yy, xx: torch.gather(rows, 1, (yy * W + xx).unsqueeze(-1).expand(-1, -1, C))
    wa = (x1.float() - x) * (y1.float() - y)
    wb = (x1.float() - x) * (y - y0.float())
    wc = (x - x0.float()) * (y1.float() - y)
    wd = (x - x0.float()) * (y - y0.float())
    out = take(y0, x0) * wa.unsqueeze(-1) + take(y1, x0) * wb.unsqueeze(-1) + take(y0, x1) * wc.unsqueeze(-1) \
        + take(y1, x1) * wd.unsqueeze(-1)
    return out.view(B, R, grid * grid, C), pts


class FoldedMLP:
    """Eval-mode stack of (Conv1d(k=1) | Linear) + BatchNorm1d + ReLU [+ Dropout] ... as plain fp32
    `relu(x @ W' + b')` launches with the norm folded into W', b'."""

    def __init__(self, layers):
        mods = [m for m in layers if not isinstance(m, nn.Dropout)]
        self.steps, i = [], 0
        while i < len(mods):
            m = mods[i]
            assert isinstance(m, (nn.Conv1d, nn.Linear)), type(m).__name__
            w = m.weight.detach().double().reshape(m.weight.shape[0], -1)
            b = m.bias.detach().double() if m.bias is not None else torch.zeros(w.shape[0], dtype=torch.float64, device=w.device)
            relu, j = False, i + 1
            if j < len(mods) and isinstance(mods[j], nn.modules.batchnorm._BatchNorm):
                bn = mods[j]
                assert not bn.training
                sc = bn.weight.detach().double() / torch.sqrt(bn.running_var.double() + bn.eps)
                w, b = w * sc[:, None], (b - bn.running_mean.double()) * sc + bn.bias.detach().double()
                j += 1
            if j < len(mods) and isinstance(mods[j], nn.ReLU):
                relu, j = True, j + 1
            self.steps.append((w.t().float().contiguous(), b.float().contiguous(), relu))
            i = j

    def __call__(self, x):
        for wt, b, relu in self.steps:
            x = torch.addmm(b, x, wt)
            if relu:
                x = torch.relu_(x)
        return x


class SparseBlockConv2d(nn.Module):
    """spconv `SparseConv2d(cin, cout, kernel_size=s, stride=s, bias=True)` as the sparse laterals of
    BEVStrideFeature use it (bev_interpolation.py:211-218): non-overlapping s x s blocks, an output cell is active
    iff one of its s * s input cells is.  Weight layout [cout, s, s, cin] (spconv 2.x), as in a checkpoint.

    forward(SparseConvTensor) -> (y [B, cout, H/s, W/s] with the conv + bias at EVERY cell, active [B, H/s, W/s]
    bool).  The block sum is a dense stride-s convolution of the densified input -- inactive cells are zeros and
    add nothing -- so it runs on the dense conv kernels; the caller keeps the active cells only."""

    def __init__(self, in_channels, out_channels, block, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.block = in_channels, out_channels, int(block)
        self.weight = nn.Parameter(torch.empty(out_channels, self.block, self.block, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if bias:
            bound = 1.0 / (self.block * self.block * in_channels) ** 0.5
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        s = self.block
        H, W = x.spatial_shape
        if H % s or W % s:
            raise NotImplementedError("SparseBlockConv2d: the grid must be a multiple of the block")
        x = x.exact()                                       # (one host wait: the active cells as an index list)
        idx = x.indices_cap.long()
        active = torch.zeros((x.batch_size, H // s, W // s), dtype=torch.bool, device=idx.device)
        active[idx[:, 0], idx[:, 1] // s, idx[:, 2] // s] = True
        dense = x.dense().to(self.weight.dtype)
        y = Fn.conv2d(dense, self.weight.permute(0, 3, 1, 2), self.bias, stride=s)
        return y, active


class SparseLateral(nn.Sequential):
    """SparseSequential(SparseConv2d(k = stride = s, bias), BatchNorm1d, ReLU) followed by `.dense()`
    (bev_interpolation.py:211-218, 277-285); children '0', '1', '2' carry the reference's state-dict keys.
    BatchNorm1d sees the active output cells only, as it does on spconv's feature rows."""

    def __init__(self, in_channels, out_channels, block):
        super().__init__(SparseBlockConv2d(in_channels, out_channels, block, bias=True),
                         build_norm_layer(_BN1, out_channels)[1], nn.ReLU())

    def forward(self, x):
        y, active = self[0](x)
        rows = y.permute(0, 2, 3, 1)[active]                # [n_active, cout]
        rows = self[2](self[1](rows)) if rows.shape[0] > 0 else rows
        out = y.new_zeros(y.shape).permute(0, 2, 3, 1)
        out[active] = rows.to(out.dtype)
        return out.permute(0, 3, 1, 2)


@SECOND_STAGE.register_module
class BEVStrideFeature(nn.Module):
    """bev_interpolation.py:162-308: fuse the neck's finest map with backbone stages at `out_stride`,
    then pool a grid_size x grid_size lattice per RoI."""

    def __init__(self, feature_sources, pillar_size, pc_range, out_stride=4, grid_size=7, in_channels=128,
                 share_channels=64, backbone_channels=None, backbone_strides=None):
        super().__init__()
        self.pillar_size, self.point_cloud_range, self.grid_size = pillar_size, pc_range, grid_size
        strides, names, chans = [1, 2, 4], ["conv1", "conv2", "conv3"], [32, 64, 128]
        assert out_stride in strides
        k = strides.index(out_stride)
        out_channels = chans[k]
        assert out_channels <= backbone_channels[names[k]]
        s = int(backbone_strides["conv3"] / out_stride)
        self.top_down_conv = nn.Sequential(nn.ConvTranspose2d(in_channels, out_channels, s, stride=s, bias=False),
                                           build_norm_layer(_BN2, out_channels)[1], nn.ReLU())
        c_in = out_channels
        self.lat_conv, self.lat_conv_name, self.lat_tensor_type = nn.ModuleList(), [], []
        for src in feature_sources:
            if src not in ("conv1", "conv2", "conv3", "conv4"):
                continue
            s = backbone_strides[src] / out_stride
            if s < 1:          # a stage finer than the fused map: block-summing sparse conv (bev_interpolation.py:211-218)
                self.lat_conv.append(SparseLateral(backbone_channels[src], out_channels, int(np.round(1 / s))))
                self.lat_conv_name.append(src)
                self.lat_tensor_type.append("sparse")
                c_in += out_channels
                continue
            s = int(s)
            self.lat_conv.append(nn.Sequential(
                nn.ConvTranspose2d(backbone_channels[src], out_channels, s, stride=s, bias=False),
                build_norm_layer(_BN2, out_channels)[1], nn.ReLU()))
            self.lat_conv_name.append(src)
            self.lat_tensor_type.append("dense")
            c_in += out_channels
        self.fusion_conv = nn.Sequential(nn.Conv2d(c_in, share_channels, 3, stride=1, padding=1, bias=True),
                                         build_norm_layer(dict(type="BN", eps=1e-3, momentum=0.01), share_channels)[1],
                                         nn.ReLU())
        self.out_stride = out_stride

    def _compile(self, _self=None):
        td = dense_hip.compile_stack(self.top_down_conv)
        lat = [dense_hip.compile_stack(m) if t == "dense" else None for m, t in zip(self.lat_conv, self.lat_tensor_type)]
        return td, lat, dense_hip.compile_stack(self.fusion_conv)

    def fused_rows_hip(self, example):
        """bf16 on the gather-GEMM: the up/lateral maps land in channel slices of one buffer (no concat),
        the 3x3 fusion conv + bias + BN + ReLU is one launch.  -> Rows [B*H*W, share_channels] bf16."""
        td, lat, fusion = dense_hip.PLANS.get(self, self._compile)
        x = dense_hip.Rows.from_nchw(example["bev_feature"])
        widths = [td[-1].cout] + [l[-1].cout if l is not None else m[0].out_channels for l, m in zip(lat, self.lat_conv)]
        up = 2 ** sum(isinstance(s, dense_hip.UpStep) for s in td)
        B, H, W = x.B, x.H * up, x.W * up
        cat = torch.empty((B * H * W, sum(widths)), dtype=torch.bfloat16, device=x.t.device)
        dense_hip.run_stack(td, x, cat[:, :widths[0]])
        c0 = widths[0]
        for k, src in enumerate(self.lat_conv_name):
            f = example["backbone_features"][src]
            if lat[k] is None:       # sparse lateral: block conv on the dense kernels, rows copied into the slice
                y = self.lat_conv[k](f)
                cat[:, c0:c0 + widths[k + 1]] = dense_hip.Rows.from_nchw(y.to(torch.bfloat16)).t
                c0 += widths[k + 1]
                continue
            if isinstance(f, SparseConvTensor):
                f = f.dense()
            dense_hip.run_stack(lat[k], dense_hip.Rows.from_nchw(f.to(torch.bfloat16)), cat[:, c0:c0 + widths[k + 1]])
            c0 += widths[k + 1]
        return dense_hip.run_stack(fusion, dense_hip.Rows(cat, B, H, W))

    def _hip_plan_fits(self):
        """the gather-GEMM plans hold 1x1 and 2x2 stride-2 transposed convs; coarser up-sampling stays on torch."""
        convs = [m for m in self.modules() if isinstance(m, (nn.ConvTranspose2d, nn.Conv2d))]
        return all(m.kernel_size in ((1, 1), (2, 2)) for m in convs if isinstance(m, nn.ConvTranspose2d)) and \
            all(m.in_channels % 32 == 0 and m.out_channels % 4 == 0 for m in convs)     # the kernels' channel granularity

    def fused_map(self, example):
        if getattr(self, "_dense_hip", False) and example["bev_feature"].dtype == torch.bfloat16 and self._hip_plan_fits():
            return self.fused_rows_hip(example).nchw()
        ref = next(self.parameters())
        cast = lambda t: t if t.dtype == ref.dtype else t.to(ref.dtype)
        feats = [self.top_down_conv(cast(example["bev_feature"]))]
        for k, src in enumerate(self.lat_conv_name):
            f = example["backbone_features"][src]
            if self.lat_tensor_type[k] == "sparse":
                feats.append(cast(self.lat_conv[k](f)))
                continue
            if isinstance(f, SparseConvTensor):
                f = f.dense()
            feats.append(self.lat_conv[k](cast(f)))
        return self.fusion_conv(torch.cat(feats, dim=1))

    def forward(self, example):
        B, R = example["rois"].shape[:2]
        fmap = self.fused_map(example)
        if fmap.requires_grad:      # training: the same kernel forward, the map's gradient through its HIP backward
            pooled, pts = roi_grid_pool_train(example["rois"].detach(), fmap, self.grid_size,
                                              self.out_stride * self.pillar_size,
                                              self.point_cloud_range[0], self.point_cloud_range[1])
        else:
            r = dense_hip.Rows.from_nchw(fmap)
            pooled, pts = roi_grid_pool_rows(example["rois"], r.t, r.B, r.H, r.W, self.grid_size,
                                             self.out_stride * self.pillar_size, self.point_cloud_range[0],
                                             self.point_cloud_range[1])
        example["roi_features"] = pooled.view(B, R, -1)
        example["point_features"] = pooled
        example["point_coords"] = pts
        return example


@SECOND_STAGE.register_module
class BEVFeature(BEVStrideFeature):
    """bev_interpolation.py:17-159: the older fusion module -- the neck map comes from the stride of `conv4`
    (BEVStrideFeature: of `conv3`), `conv4` itself may be a lateral source, and a source AT the output stride takes a
    1x1 sparse conv unless the output stride is 8 (bev_interpolation.py:56).  Map fusion, the rotated RoI lattice
    and the bilinear gather are the parent's (same forward: bev_interpolation.py:125-158 against :270-308)."""

    def __init__(self, feature_sources, pillar_size, pc_range, out_stride=4, grid_size=7, in_channels=256,
                 share_channels=64, backbone_channels=None, backbone_strides=None):
        nn.Module.__init__(self)
        self.pillar_size, self.point_cloud_range, self.grid_size = pillar_size, pc_range, grid_size
        strides, names, chans = [1, 2, 4, 8], ["conv1", "conv2", "conv3", "conv4"], [32, 64, 128, 256]
        assert out_stride in strides
        k = strides.index(out_stride)
        out_channels = chans[k]
        assert out_channels <= backbone_channels[names[k]]
        s = int(backbone_strides["conv4"] / out_stride)
        self.top_down_conv = nn.Sequential(nn.ConvTranspose2d(in_channels, out_channels, s, stride=s, bias=False),
                                           build_norm_layer(_BN2, out_channels)[1], nn.ReLU())
        c_in = out_channels
        self.lat_conv, self.lat_conv_name, self.lat_tensor_type = nn.ModuleList(), [], []
        for src in feature_sources:
            if src not in names:
                continue
            s = backbone_strides[src] / out_stride
            if s > 1 or (out_stride == 8 and s == 1):
                s = int(s)
                self.lat_conv.append(nn.Sequential(
                    nn.ConvTranspose2d(backbone_channels[src], out_channels, s, stride=s, bias=False),
                    build_norm_layer(_BN2, out_channels)[1], nn.ReLU()))
                self.lat_tensor_type.append("dense")
            else:
                self.lat_conv.append(SparseLateral(backbone_channels[src], out_channels, int(np.round(1 / s))))
                self.lat_tensor_type.append("sparse")
            self.lat_conv_name.append(src)
            c_in += out_channels
        self.fusion_conv = nn.Sequential(nn.Conv2d(c_in, share_channels, 3, stride=1, padding=1, bias=True),
                                         build_norm_layer(_BN2, share_channels)[1], nn.ReLU())
        self.out_stride = out_stride


def rotate_points_along_z(points, angle):
    """box_torch_ops.py:120-138."""
    c, s = torch.cos(angle), torch.sin(angle)
    z, o = angle.new_zeros(points.shape[0]), angle.new_ones(points.shape[0])
    rot = torch.stack((c, -s, z, s, c, z, z, z, o), dim=1).view(-1, 3, 3).float()
    return torch.cat((torch.matmul(points[:, :, 0:3], rot), points[:, :, 3:]), dim=-1)


def limit_period(val, offset=0.5, period=np.pi):
    """roi_head_template.py:9-10."""
    return val - torch.floor(val / period + offset) * period


def points_in_boxes_2d(points, boxes):
    """roiaware_pool3d_utils.py:58-81: points [B, M, 2], boxes [B, T, 7] (det3d convention) -> [B, M] int32,
    index of the first box holding the point in the BEV plane, else -1.  One HIP launch."""
    if points.device.type != "cuda":
        raise _lib.PillarnetHipError("points_in_boxes_2d: tensors must live on the GPU (no CPU fallback)")
    assert boxes.shape[0] == points.shape[0] and boxes.shape[2] == 7 and points.shape[2] == 2
    B, M, _ = points.shape
    bx = iou3d_nms.to_pcdet(boxes.reshape(-1, 7)).reshape(boxes.shape).contiguous().float()
    pts = points.contiguous().float()
    out = torch.empty((B, M), dtype=torch.int32, device=points.device)
    rc = _lib.lib().pn_points_in_boxes_2d(_lib.ptr(bx), _lib.ptr(pts), B, boxes.shape[1], M, _lib.ptr(out),
                                          _lib.current_stream(points.device))
    _lib.check(rc, "pn_points_in_boxes_2d")
    return out


class ProposalTargetLayer(nn.Module):
    """roi_heads/target_assigner/proposal_target_layer.py:8-232.  The sampling draws the reference's random
    numbers in the reference's order (np.random.permutation / np.random.rand for foreground, CPU torch.randint
    for background), so equal seeds give equal samples.  `iou_fn` defaults to the HIP rotated 3-D IoU."""

    def __init__(self, roi_sampler_cfg, iou_fn=None):
        super().__init__()
        self.cfg = roi_sampler_cfg
        self.iou_fn = iou_fn or iou3d_nms.boxes_iou3d_gpu

    def forward(self, batch_dict):
        c = self.cfg
        rois, gt_of_rois, ious, scores, labels = self.sample_rois_for_rcnn(batch_dict)
        reg_valid_mask = (ious > c["REG_FG_THRESH"]).long()
        if c["CLS_SCORE_TYPE"] == "cls":
            cls_labels = (ious > c["CLS_FG_THRESH"]).long()
            cls_labels[(ious > c["CLS_BG_THRESH"]) & (ious < c["CLS_FG_THRESH"])] = -1
        elif c["CLS_SCORE_TYPE"] == "roi_iou":
            bg, fg = c["CLS_BG_THRESH"], c["CLS_FG_THRESH"]
            fg_mask, bg_mask = ious > fg, ious < bg
            interval = (fg_mask == 0) & (bg_mask == 0)
            cls_labels = (fg_mask > 0).float()
            cls_labels[interval] = (ious[interval] - bg) / (fg - bg)
        else:
            raise NotImplementedError(c["CLS_SCORE_TYPE"])
        return {"rois": rois, "gt_of_rois": gt_of_rois, "gt_iou_of_rois": ious, "roi_scores": scores,
                "roi_labels": labels, "reg_valid_mask": reg_valid_mask, "rcnn_cls_labels": cls_labels}

    def sample_rois_for_rcnn(self, batch_dict):
        B, n = batch_dict["batch_size"], self.cfg["ROI_PER_IMAGE"]
        rois, roi_scores, roi_labels = batch_dict["rois"], batch_dict["roi_scores"], batch_dict["roi_labels"]
        gt_boxes = batch_dict["gt_boxes_and_cls"]
        code = rois.shape[-1]
        out_rois, out_gt = rois.new_zeros(B, n, code), rois.new_zeros(B, n, code + 1)
        out_iou, out_scores = rois.new_zeros(B, n), rois.new_zeros(B, n)
        out_labels = rois.new_zeros((B, n), dtype=torch.long)
        for i in range(B):
            cur_gt = gt_boxes[i]
            k = len(cur_gt) - 1
            while k > 0 and cur_gt[k].sum() == 0:
                k -= 1
            cur_gt = cur_gt[: k + 1]
            if len(cur_gt) == 0:
                cur_gt = cur_gt.new_zeros((1, cur_gt.shape[1]))
            if self.cfg.get("SAMPLE_ROI_BY_EACH_CLASS", False):
                max_ov, assign = self.get_max_iou_with_same_class(rois[i][:, :7], roi_labels[i], cur_gt[:, 0:7],
                                                                  cur_gt[:, -1].long())
            else:
                max_ov, assign = torch.max(self.iou_fn(rois[i], cur_gt[:, 0:7]), dim=1)
            idx = self.subsample_rois(max_ov)
            out_rois[i], out_labels[i], out_iou[i] = rois[i][idx], roi_labels[i][idx], max_ov[idx]
            out_scores[i], out_gt[i] = roi_scores[i][idx], cur_gt[assign[idx]]
        return out_rois, out_gt, out_iou, out_scores, out_labels

    def subsample_rois(self, max_overlaps):
        c = self.cfg
        n = c["ROI_PER_IMAGE"]
        fg_per_image = int(np.round(c["FG_RATIO"] * n))
        fg_thresh = min(c["REG_FG_THRESH"], c["CLS_FG_THRESH"])
        fg_inds = (max_overlaps >= fg_thresh).nonzero().view(-1)
        easy_bg = (max_overlaps < c["CLS_BG_THRESH_LO"]).nonzero().view(-1)
        hard_bg = ((max_overlaps < c["REG_FG_THRESH"]) & (max_overlaps >= c["CLS_BG_THRESH_LO"])).nonzero().view(-1)
        n_fg, n_bg = fg_inds.numel(), hard_bg.numel() + easy_bg.numel()
        dev = max_overlaps.device
        if n_fg > 0 and n_bg > 0:
            take = min(fg_per_image, n_fg)
            perm = torch.from_numpy(np.random.permutation(n_fg)).long().to(dev)
            fg_inds = fg_inds[perm[:take]]
            bg_inds = self.sample_bg_inds(hard_bg, easy_bg, n - take, c["HARD_BG_RATIO"])
        elif n_fg > 0 and n_bg == 0:
            pick = torch.from_numpy(np.floor(np.random.rand(n) * n_fg)).long().to(dev)
            fg_inds = fg_inds[pick]
            bg_inds = fg_inds.new_zeros(0)
        elif n_bg > 0 and n_fg == 0:
            bg_inds = self.sample_bg_inds(hard_bg, easy_bg, n, c["HARD_BG_RATIO"])
        else:
            raise NotImplementedError("no foreground and no background RoI")
        return torch.cat((fg_inds, bg_inds), dim=0)

    @staticmethod
    def sample_bg_inds(hard_bg, easy_bg, n_bg, hard_ratio):
        def draw(pool, count):          # CPU generator, as the reference's device-less torch.randint
            return pool[torch.randint(low=0, high=pool.numel(), size=(count,)).long().to(pool.device)]
        if hard_bg.numel() > 0 and easy_bg.numel() > 0:
            n_hard = min(int(n_bg * hard_ratio), len(hard_bg))
            h = draw(hard_bg, n_hard)
            return torch.cat([h, draw(easy_bg, n_bg - n_hard)], dim=0)
        if hard_bg.numel() > 0:
            return draw(hard_bg, n_bg)
        if easy_bg.numel() > 0:
            return draw(easy_bg, n_bg)
        raise NotImplementedError

    def get_max_iou_with_same_class(self, rois, roi_labels, gt_boxes, gt_labels):
        max_ov = rois.new_zeros(rois.shape[0])
        assign = roi_labels.new_zeros(roi_labels.shape[0])
        for k in range(gt_labels.min().item(), gt_labels.max().item() + 1):
            roi_mask, gt_mask = roi_labels == k, gt_labels == k
            if roi_mask.sum() > 0 and gt_mask.sum() > 0:
                orig = gt_mask.nonzero().view(-1)
                cur_max, cur_assign = torch.max(self.iou_fn(rois[roi_mask], gt_boxes[gt_mask]), dim=1)
                max_ov[roi_mask] = cur_max
                assign[roi_mask] = orig[cur_assign]
        return max_ov, assign


def sigmoid_focal_loss(logits, target, weights, alpha=0.25, gamma=2.0):
    """loss_utils.py:10-75 SigmoidFocalClassificationLoss.forward."""
    p = torch.sigmoid(logits)
    alpha_w = target * alpha + (1 - target) * (1 - alpha)
    pt = target * (1.0 - p) + (1.0 - target) * p
    bce = torch.clamp(logits, min=0) - logits * target + torch.log1p(torch.exp(-torch.abs(logits)))
    loss = alpha_w * torch.pow(pt, gamma) * bce
    if weights.dim() == 2 or (weights.dim() == 1 and target.dim() == 2):
        weights = weights.unsqueeze(-1)
    assert weights.dim() == loss.dim()
    return loss * weights


class _Swap(nn.Module):
    """(b, n, d) <-> (b, d, n): the parameter-free `Rearrange` steps of mlp_layers.py:89-93."""

    def forward(self, x):
        return x.transpose(1, 2)


class MLPMixer(nn.Module):
    """roi_heads/mlp_layers.py:24-63: token mixing over the RoI grid points, then channel mixing, both residual."""

    def __init__(self, in_channels, num_patches, expansion_factor=4, expansion_factor_token=0.5):
        super().__init__()
        inner = int(num_patches * expansion_factor)
        self.token_mixer = nn.Sequential(build_norm_layer(dict(type="LN"), in_channels)[1],
                                         nn.Conv1d(num_patches, inner, kernel_size=1), nn.GELU(),
                                         nn.Conv1d(inner, num_patches, kernel_size=1))
        inner = int(in_channels * expansion_factor_token)
        self.channel_mixer = nn.Sequential(build_norm_layer(dict(type="LN"), in_channels)[1],
                                           nn.Linear(in_channels, inner), nn.GELU(), nn.Linear(inner, in_channels))

    def forward(self, x):                     # [B, patches, C]
        x = self.token_mixer(x) + x
        return self.channel_mixer(x) + x


class _Affine(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones((1, 1, dim)))
        self.beta = nn.Parameter(torch.zeros((1, 1, dim)))

    def forward(self, x):
        return self.alpha * x + self.beta


class ResMLPLayer(nn.Module):
    """roi_heads/mlp_layers.py:66-117."""

    def __init__(self, in_channels, num_patches, expansion_factor=2, layer_scale_init=1e-4):
        super().__init__()
        self.token_aff = _Affine(in_channels)
        self.token_scale = nn.Parameter(layer_scale_init * torch.ones(in_channels))
        self.token_mixer = nn.Sequential(_Swap(), nn.Linear(num_patches, num_patches), _Swap())
        self.channel_aff = _Affine(in_channels)
        self.channel_scale = nn.Parameter(layer_scale_init * torch.ones(in_channels))
        self.channel_mixer = nn.Sequential(nn.Linear(in_channels, in_channels * expansion_factor), nn.GELU(),
                                           nn.Linear(in_channels * expansion_factor, in_channels))
        self.post_aff = _Affine(in_channels)

    def forward(self, x):
        x = self.token_aff(x)
        x = x + self.token_scale * self.token_mixer(x)
        x = self.channel_aff(x)
        x = x + self.channel_scale * self.channel_mixer(x)
        return self.post_aff(x)


def _fc_stack(cfg_list, cin, cout, dp_ratio, conv=True):
    layers, pre = [], cin
    for k, c in enumerate(cfg_list):
        layers += [nn.Conv1d(pre, c, kernel_size=1, bias=False), build_norm_layer(_BN1, c)[1], nn.ReLU()]
        pre = c
        if dp_ratio >= 0 and k == 0:
            layers.append(nn.Dropout(dp_ratio))
    layers.append(nn.Conv1d(pre, cout, kernel_size=1, bias=True))
    return nn.Sequential(*layers)


class RoIHeadTemplate(nn.Module):
    """roi_head_template.py:13-219: what every RoI head shares -- the proposal sampler, the RoI-frame target
    encoding, class / box / IoU losses and the decoding of residuals back into boxes."""

    def __init__(self, num_class, model_cfg):
        super().__init__()
        self.model_cfg, self.num_class = model_cfg, num_class
        self.proposal_target_layer = ProposalTargetLayer(model_cfg["TARGET_CONFIG"]) if "TARGET_CONFIG" in model_cfg else None
        self.use_iou, self.forward_ret_dict = False, None

    def make_fc_layers(self, input_channels, output_channels, fc_list):
        """roi_head_template.py:23-39."""
        return _fc_stack(list(fc_list), input_channels, output_channels, self.model_cfg["DP_RATIO"])

    def generate_predicted_boxes(self, batch_size, rois, cls_preds, box_preds):
        """roi_head_template.py:189-219."""
        code = box_preds.shape[-1]
        cls = cls_preds.view(batch_size, -1, cls_preds.shape[-1])
        ry, xyz = rois[:, :, 6].reshape(-1), rois[:, :, 0:3].reshape(-1, 3)
        local = rois.clone().detach()
        local[:, :, 0:3] = 0
        boxes = (box_preds.view(batch_size, -1, code) + local).view(-1, code)
        boxes = rotate_points_along_z(boxes.unsqueeze(1), ry).squeeze(1)
        boxes[:, 0:3] += xyz
        return cls, boxes.view(batch_size, -1, code)

    def assign_targets(self, batch_dict):
        """roi_head_template.py:41-83: sample RoIs, then express each RoI's GT box in the RoI's own frame."""
        B = batch_dict["batch_size"]
        with torch.no_grad():
            td = self.proposal_target_layer(batch_dict)
        rois, gt = td["rois"], td["gt_of_rois"]
        td["gt_of_rois_src"] = gt.clone().detach()
        roi_ry = limit_period(rois[:, :, 6], offset=0.5, period=np.pi * 2)
        gt[:, :, :6] = gt[:, :, :6] - rois[:, :, :6]
        gt[:, :, 6] = gt[:, :, 6] - roi_ry
        gt = rotate_points_along_z(gt.view(-1, 1, gt.shape[-1]), -roi_ry.view(-1)).view(B, -1, gt.shape[-1])
        if rois.shape[-1] == 9:
            gt[:, :, 7:-1] = gt[:, :, 7:-1] - rois[:, :, 7:]
        heading = gt[:, :, 6] % (2 * np.pi)
        opposite = (heading > np.pi * 0.5) & (heading < np.pi * 1.5)
        heading[opposite] = (heading[opposite] + np.pi) % (2 * np.pi)
        flag = heading > np.pi
        heading[flag] = heading[flag] - np.pi * 2
        gt[:, :, 6] = torch.clamp(heading, min=-np.pi / 2, max=np.pi / 2)
        td["gt_of_rois"] = gt
        return td

    def get_box_cls_layer_loss(self, ret):
        """roi_head_template.py:149-168."""
        lc = self.model_cfg["LOSS_CONFIG"]
        labels = ret["rcnn_cls_labels"].view(-1)
        valid = (labels >= 0).float()
        if lc["CLS_LOSS"] == "BinaryCrossEntropy":
            per = Fn.binary_cross_entropy(torch.sigmoid(ret["rcnn_cls"].view(-1)), labels.float(), reduction="none")
        elif lc["CLS_LOSS"] == "CrossEntropy":
            per = Fn.cross_entropy(ret["rcnn_cls"], labels, reduction="none", ignore_index=-1)
        else:
            raise NotImplementedError(lc["CLS_LOSS"])
        loss = (per * valid).sum() / torch.clamp(valid.sum(), min=1.0) * lc["LOSS_WEIGHTS"]["rcnn_cls_weight"]
        return loss, {"rcnn_loss_cls": loss.detach()}

    def get_box_reg_layer_loss(self, ret):
        """roi_head_template.py:85-115."""
        lc = self.model_cfg["LOSS_CONFIG"]
        if lc["REG_LOSS"] != "L1":
            raise NotImplementedError(lc["REG_LOSS"])
        code = ret["rcnn_reg"].shape[-1]
        fg = ret["reg_valid_mask"].view(-1) > 0
        target = ret["gt_of_rois"][..., 0:code].reshape(-1, code)
        n = target.shape[0]
        per = Fn.l1_loss(ret["rcnn_reg"].view(n, -1), target, reduction="none")
        per = per * per.new_tensor(lc["LOSS_WEIGHTS"]["code_weights"])
        loss = (per.view(n, -1) * fg.unsqueeze(-1).float()).sum() / max(int(fg.long().sum().item()), 1)
        loss = loss * lc["LOSS_WEIGHTS"]["rcnn_reg_weight"]
        return loss, {"rcnn_loss_reg": loss.detach()}

    def get_box_iou_layer_loss(self, ret):
        """roi_head_template.py:117-147: the IoU branch regresses 2 * IoU3D(decoded box, GT) - 1 on the foreground
        RoIs (rotated 3-D IoU on the HIP kernel, det3d/ops/iou3d_nms/iou3d_nms_utils.py:77-111)."""
        lc = self.model_cfg["LOSS_CONFIG"]
        if lc["IOU_LOSS"] != "L1":
            raise NotImplementedError(lc["IOU_LOSS"])
        fg = ret["reg_valid_mask"].view(-1) > 0
        n_fg = int(fg.long().sum().item())
        iou_fn = getattr(self, "aligned_iou_fn", None) or iou3d_nms.boxes_aligned_iou3d_gpu
        pred = ret["batch_box_preds"].view(-1, 7).detach()[fg]
        gt = ret["gt_of_rois_src"][..., :7].reshape(-1, 7)[fg]
        target = 2 * iou_fn(pred, gt) - 1 if n_fg else pred.new_zeros((0, 1))
        # shapes as in the reference: predictions [n] against targets [n, 1] broadcast to ALL n x n pairs before the
        # sum (roi_head_template.py:131-139) -- kept, so that losses and gradients equal the reference's
        est = ret["rcnn_iou"].view(-1)[fg]
        loss = (est.view(1, -1) - target.view(-1, 1).to(est.dtype)).abs().sum()
        loss = loss / max(n_fg, 1) * lc["LOSS_WEIGHTS"]["rcnn_iou_weight"]
        return loss, {"rcnn_loss_iou": loss.detach()}

    def get_loss(self, tb_dict=None):
        """roi_head_template.py:170-187."""
        tb = {} if tb_dict is None else tb_dict
        total, t = self.get_box_cls_layer_loss(self.forward_ret_dict)
        tb.update(t)
        reg, t = self.get_box_reg_layer_loss(self.forward_ret_dict)
        total = total + reg
        tb.update(t)
        if self.use_iou:
            iou, t = self.get_box_iou_layer_loss(self.forward_ret_dict)
            total = total + iou
            tb.update(t)
        tb["rcnn_loss"] = total.item()
        return total, tb


@ROI_HEAD.register_module
class RoIMIXHead(RoIHeadTemplate):
    """roi_mix_head.py:17-122 on the shared template (roi_head_template.py)."""

    def __init__(self, in_channels=None, model_cfg=None, num_class=1, code_size=7, add_box_param=False, test_cfg=None,
                 mixer_type=None, num_patches=49, **kwargs):
        super().__init__(num_class=num_class, model_cfg=model_cfg)
        if in_channels is None:
            in_channels = kwargs.pop("input_channels")          # the legacy configs' name for it
        self.test_cfg = test_cfg
        self.code_size, self.add_box_param, self.num_patches = code_size, add_box_param, num_patches
        if mixer_type == "MLPMixer":
            self.mlp_mixer = MLPMixer(in_channels=in_channels, num_patches=num_patches)
        elif mixer_type == "ResMLP":
            self.mlp_mixer = ResMLPLayer(in_channels=in_channels, num_patches=num_patches)
        else:
            self.mlp_mixer = nn.Sequential()
        pre = in_channels * num_patches
        shared, fcs = [], list(model_cfg["SHARED_FC"])
        for k, c in enumerate(fcs):
            shared += [nn.Conv1d(pre, c, kernel_size=1, bias=False), build_norm_layer(_BN1, c)[1], nn.ReLU()]
            pre = c
            if k != len(fcs) - 1 and model_cfg["DP_RATIO"] > 0:
                shared.append(nn.Dropout(model_cfg["DP_RATIO"]))
        self.shared_fc_layer = nn.Sequential(*shared)
        self.cls_layers = self.make_fc_layers(pre, num_class, model_cfg["CLS_FC"])
        self.reg_layers = self.make_fc_layers(pre, code_size, model_cfg["REG_FC"])
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.reg_layers[-1].weight, mean=0, std=0.001)

    def _forward_train(self, batch_dict):
        """roi_mix_head.py:81-121 with training=True: the module path (BatchNorm statistics, Dropout)."""
        td = batch_dict.get("roi_targets_dict", None)
        if td is None:
            td = self.assign_targets(batch_dict)
            batch_dict["rois"], batch_dict["roi_labels"], batch_dict["roi_scores"] = td["rois"], td["roi_labels"], td["roi_scores"]
        B, R = batch_dict["rois"].shape[:2]
        feats = self.mlp_mixer(batch_dict["roi_features"].view(B * R, self.num_patches, -1))
        if self.add_box_param:
            feats = torch.cat([feats, batch_dict["rois"], batch_dict["roi_scores"].unsqueeze(-1)], dim=-1)
        x = self.shared_fc_layer(feats.reshape(B * R, -1, 1))
        td["rcnn_cls"] = self.cls_layers(x).transpose(1, 2).contiguous().squeeze(dim=1)
        td["rcnn_reg"] = self.reg_layers(x).transpose(1, 2).contiguous().squeeze(dim=1)
        self.forward_ret_dict = td
        return batch_dict

    def forward(self, batch_dict, training=True):
        if training:
            return self._forward_train(batch_dict)
        B = len(batch_dict["rois"])
        R = batch_dict["rois"].shape[1]
        feats = self.mlp_mixer(batch_dict["roi_features"].view(B * R, self.num_patches, -1))
        if self.add_box_param:
            feats = torch.cat([feats, batch_dict["rois"], batch_dict["roi_scores"].unsqueeze(-1)], dim=-1)
        shared, cls_mlp, reg_mlp = dense_hip.PLANS.get(self, lambda m: (
            FoldedMLP(m.shared_fc_layer), FoldedMLP(m.cls_layers), FoldedMLP(m.reg_layers)))
        x = shared(feats.reshape(B * R, -1).float())
        cls, reg = cls_mlp(x), reg_mlp(x)
        c, b = self.generate_predicted_boxes(batch_dict["batch_size"], batch_dict["rois"], cls, reg)
        batch_dict["batch_cls_preds"], batch_dict["batch_box_preds"] = c, b
        batch_dict["cls_preds_normalized"] = False
        return batch_dict


class FFN(nn.Module):
    """mmcv.cnn.bricks.transformer.FFN as roi_ffn_head.py:34-39 constructs it (mmcv is a third-party package that
    is not installed here: restated from its published definition).  num_fcs Linear layers -- the first
    num_fcs - 1 widen to `feedforward_channels` with activation + dropout -- and the input added back.
    Parameter names as in mmcv: layers.<i>.0.{weight,bias} for the widening layers, layers.<num_fcs-1>.{weight,bias}
    for the last."""

    def __init__(self, embed_dims, feedforward_channels, num_fcs=2, act_cfg=None, dropout=0.0, add_residual=True):
        super().__init__()
        assert num_fcs >= 2, f"num_fcs should be no less than 2. got {num_fcs}."
        act = dict(act_cfg or dict(type="ReLU", inplace=True))
        kind = act.pop("type")
        if kind not in ("ReLU", "GELU"):
            raise NotImplementedError(f"FFN activation {kind}")
        layers, c = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(c, feedforward_channels),
                                        nn.ReLU(**act) if kind == "ReLU" else nn.GELU(), nn.Dropout(dropout)))
            c = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        self.layers = nn.Sequential(*layers)
        self.dropout = nn.Dropout(dropout)
        self.add_residual = add_residual

    def forward(self, x, residual=None):
        out = self.dropout(self.layers(x))
        if not self.add_residual:
            return out
        return (x if residual is None else residual) + out


@ROI_HEAD.register_module
class RoIFFNHead(RoIHeadTemplate):
    """roi_ffn_head.py:10-140: a transformer-style feed-forward block + LayerNorm over the flattened RoI feature,
    then Linear + LayerNorm + ReLU towers for class, box residual and (num_iou_fcs > 0) an IoU estimate."""

    def __init__(self, model_cfg, num_cls_fcs=1, num_reg_fcs=1, num_iou_fcs=1, feedforward_channels=2048,
                 content_dim=256, num_ffn_fcs=2, dropout=0.0, ffn_act_cfg=dict(type="ReLU", inplace=True),
                 num_class=1, code_size=7, test_cfg=None, init_bias=-2.19, add_box_param=False):
        super().__init__(num_class=num_class, model_cfg=model_cfg)
        self.test_cfg, self.code_size, self.add_box_param, self.init_bias = test_cfg, code_size, add_box_param, init_bias
        self.use_iou = num_iou_fcs > 0
        self.ffn = FFN(content_dim, feedforward_channels, num_ffn_fcs, act_cfg=ffn_act_cfg, dropout=dropout)
        self.ffn_norm = build_norm_layer(dict(type="LN"), content_dim)[1]

        def tower(n):
            mods = nn.ModuleList()
            for _ in range(n):
                mods += [nn.Linear(content_dim, content_dim, bias=True), build_norm_layer(dict(type="LN"), content_dim)[1],
                         nn.ReLU()]
            return mods

        self.cls_fcs = tower(num_cls_fcs)
        self.fc_cls = nn.Linear(content_dim, self.num_class)
        self.reg_fcs = tower(num_reg_fcs)
        self.fc_reg = nn.Linear(content_dim, code_size)
        if self.use_iou:
            self.iou_fcs = tower(num_iou_fcs)
            self.fc_iou = nn.Linear(content_dim, 1)
        self.init_weights()

    @torch.no_grad()
    def init_weights(self):
        """roi_ffn_head.py:66-79."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.reset_parameters()
                nn.init.xavier_uniform_(m.weight)
        nn.init.zeros_(self.fc_reg.weight)
        nn.init.zeros_(self.fc_reg.bias)
        nn.init.constant_(self.fc_cls.bias, self.init_bias)
        if self.use_iou:
            nn.init.zeros_(self.fc_iou.weight)
            nn.init.zeros_(self.fc_iou.bias)

    def forward(self, batch_dict, training=True):
        """roi_ffn_head.py:81-140."""
        if training:
            td = batch_dict.get("roi_targets_dict", None)
            if td is None:
                td = self.assign_targets(batch_dict)
                batch_dict["rois"], batch_dict["roi_labels"], batch_dict["roi_scores"] = td["rois"], td["roi_labels"], td["roi_scores"]
        if self.add_box_param:
            batch_dict["roi_features"] = torch.cat([batch_dict["roi_features"], batch_dict["rois"],
                                                    batch_dict["roi_scores"].unsqueeze(-1)], dim=-1)
        batch_dict["roi_features"] = batch_dict["roi_features"].view(-1, batch_dict["roi_features"].shape[-1])
        query = self.ffn_norm(self.ffn(batch_dict["roi_features"]))

        def run(layers, x):
            for m in layers:
                x = m(x)
            return x

        rcnn_cls = self.fc_cls(run(self.cls_fcs, query))
        rcnn_reg = self.fc_reg(run(self.reg_fcs, query))
        rcnn_iou = self.fc_iou(run(self.iou_fcs, query)) if self.use_iou else None
        if not training:
            c, b = self.generate_predicted_boxes(batch_dict["batch_size"], batch_dict["rois"], rcnn_cls, rcnn_reg)
            batch_dict["batch_cls_preds"], batch_dict["batch_box_preds"] = c, b
            batch_dict["cls_preds_normalized"] = False
            return batch_dict
        td["rcnn_cls"], td["rcnn_reg"] = rcnn_cls, rcnn_reg
        if self.use_iou:
            td["rcnn_iou"] = rcnn_iou
            _, td["batch_box_preds"] = self.generate_predicted_boxes(batch_dict["batch_size"], batch_dict["rois"],
                                                                     rcnn_cls, rcnn_reg)
        self.forward_ret_dict = td
        return batch_dict


@POINT_HEAD.register_module
class PointHead(nn.Module):
    """point_head_simple.py:14-98 (the auxiliary foreground classifier on the RoI grid points)."""

    def __init__(self, in_channels=None, num_class=1, model_cfg=None, **kwargs):
        super().__init__()
        if in_channels is None:
            in_channels = kwargs.pop("input_channels")          # the legacy configs' name for it
        self.model_cfg, self.num_class = model_cfg, num_class
        layers, c_in = [], in_channels
        for c in model_cfg["CLS_FC"]:
            layers += [nn.Linear(c_in, c, bias=False), build_norm_layer(_BN1, c)[1], nn.ReLU()]
            c_in = c
        layers.append(nn.Linear(c_in, 1, bias=True))
        self.cls_layers = nn.Sequential(*layers)

    def assign_targets(self, input_dict, in_boxes_fn=None):
        """point_head_simple.py:23-60 + point_head_template.py:141-170: a grid point is foreground when it lies
        inside the BEV rectangle of the GT box its RoI was matched to."""
        pts = input_dict["point_coords"]
        assert pts[0].shape[-1] == 2, "only the 2-D (RoI grid point) assignment is built"
        gt = input_dict["roi_targets_dict"]["gt_of_rois_src"]
        assert gt.dim() == 3 and gt.shape[2] == 8, f"gt_boxes.shape={tuple(gt.shape)}"
        assert self.num_class == 1
        fn = in_boxes_fn or points_in_boxes_2d
        labels = []
        for k in range(gt.shape[0]):
            p = pts[k].contiguous()                                  # [rois, grid points, 2]
            idx = fn(p, gt[k, :, 0:7].unsqueeze(dim=1).contiguous()).long()
            lab = gt.new_zeros(p.shape[0], p.shape[1]).long()
            lab[idx >= 0] = 1
            labels.append(lab.view(-1))
        return {"point_cls_labels": torch.cat(labels, dim=0)}

    def get_loss(self, tb_dict=None):
        """point_head_template.py:172-195 (sigmoid focal loss, normalised by the positives)."""
        tb = {} if tb_dict is None else tb_dict
        labels = self.forward_ret_dict["point_cls_labels"].view(-1)
        preds = self.forward_ret_dict["point_cls_preds"].view(-1, self.num_class)
        pos = labels > 0
        w = ((labels == 0) * 1.0 + 1.0 * pos).float()
        pos_norm = pos.sum(dim=0).float()
        w = w / torch.clamp(pos_norm, min=1.0)
        one_hot = preds.new_zeros(*labels.shape, self.num_class + 1)
        one_hot.scatter_(-1, (labels * (labels >= 0).long()).unsqueeze(-1).long(), 1.0)
        loss = sigmoid_focal_loss(preds, one_hot[..., 1:], w).sum()
        loss = loss * self.model_cfg["LOSS_CONFIG"]["LOSS_WEIGHTS"]["point_cls_weight"]
        tb.update(point_loss_cls=loss.item(), point_pos_num=pos_norm.item())
        return loss, tb

    def forward(self, batch_dict):
        if self.training:
            pf = batch_dict["point_features"]
            preds = self.cls_layers(pf.reshape(-1, pf.shape[-1]))
            batch_dict["point_cls_scores"] = torch.sigmoid(preds)
            if self.model_cfg.get("ATT_MODEL", False):
                batch_dict["point_features"] = pf * batch_dict["point_cls_scores"].view(*pf.shape[:-1], 1)
            self.forward_ret_dict = {"point_cls_preds": preds,
                                     "point_cls_labels": self.assign_targets(batch_dict)["point_cls_labels"]}
            return batch_dict
        pf = batch_dict["point_features"]
        mlp = dense_hip.PLANS.get(self, lambda m: FoldedMLP(m.cls_layers))
        batch_dict["point_cls_scores"] = torch.sigmoid(mlp(pf.reshape(-1, pf.shape[-1]).float()))
        if self.model_cfg.get("ATT_MODEL", False):
            batch_dict["point_features"] = pf * batch_dict["point_cls_scores"].view(*pf.shape[:-1], 1)
        return batch_dict


@DETECTORS.register_module
class PillarRCNN(BaseDetector):
    """pillar_rcnn.py:9-166."""

    def __init__(self, first_stage_cfg, second_stage_modules, roi_head, freeze=False, point_head=None, **kwargs):
        super().__init__()
        self.single_det = builder.build_detector(first_stage_cfg, **kwargs)
        if freeze:
            for p in self.single_det.parameters():
                p.requires_grad = False
        self.bbox_head = self.single_det.bbox_head
        self.test_cfg = self.single_det.test_cfg
        self.num_classes = sum(self.single_det.num_classes)
        first = dict(backbone_channels=self.single_det.backbone.backbone_channels,
                     backbone_strides=self.single_det.backbone.backbone_strides)
        self.second_stage = nn.ModuleList()
        for module in second_stage_modules:
            module = dict(module)
            module.update(first)
            self.second_stage.append(builder.build_second_stage_module(module))
        self.point_head = builder.build_point_head(point_head) if point_head is not None else None
        self.roi_head = builder.build_roi_head(roi_head)

    def to_inference(self, dtype=torch.bfloat16, device="cuda", dense_backend="hip"):
        self.single_det.to_inference(dtype, device, dense_backend)
        self.eval().to(device)
        for m in list(self.second_stage) + [self.point_head, self.roi_head]:
            if m is not None:
                m.float()      # parameters stay fp32; the RoI MLPs (< 2 GFLOP) run in fp32
        for m in self.second_stage:
            m._dense_hip = dtype == torch.bfloat16 and dense_backend == "hip"   # map fusion on the bf16 gather-GEMM
        return self

    def reorder_first_stage_prediction(self, first_pred, example):
        """pillar_rcnn.py:52-83: pad each sample's boxes to NMS_POST_MAXSIZE RoIs, labels shifted by +1."""
        B, L = len(first_pred), first_pred[0]["box3d_lidar"].shape[1]
        N = self.single_det.NMS_POST_MAXSIZE
        rois = first_pred[0]["box3d_lidar"].new_zeros((B, N, L))
        scores = first_pred[0]["scores"].new_zeros((B, N))
        labels = first_pred[0]["label_preds"].new_zeros((B, N), dtype=torch.long)
        for i, p in enumerate(first_pred):
            box = p["box3d_lidar"]
            n = box.shape[0]
            if self.roi_head.code_size == 9:
                box = box[:, [0, 1, 2, 3, 4, 5, 8, 6, 7]]
            rois[i, :n], labels[i, :n], scores[i, :n] = box, p["label_preds"] + 1, p["scores"]
        example.update(rois=rois, roi_labels=labels, roi_scores=scores, has_class_labels=True)
        return example

    def combine_loss(self, one_stage_loss, point_loss, roi_loss, tb_dict):
        """pillar_rcnn.py:44-50."""
        for _ in range(len(one_stage_loss["loss"])):
            one_stage_loss["roi_reg_loss"].append(tb_dict["rcnn_loss_reg"])
            one_stage_loss["roi_cls_loss"].append(tb_dict["rcnn_loss_cls"])
        one_stage_loss["loss"][0] = one_stage_loss["loss"][0] + roi_loss + point_loss
        return one_stage_loss

    def forward(self, example, return_loss=True, **kwargs):
        """pillar_rcnn.py:85-131."""
        example = dict(example)
        example["batch_size"] = len(example["metadata"])
        boxes, bev, backbone_features, one_stage_loss = self.single_det.forward_two_stage(example, return_loss, **kwargs)
        example["bev_feature"], example["backbone_features"] = bev[-1], backbone_features
        if self.roi_head.code_size == 7 and return_loss:
            example["gt_boxes_and_cls"] = example["gt_boxes_and_cls"][:, :, [0, 1, 2, 3, 4, 5, 6, -1]]   # drop velocity
        example = self.reorder_first_stage_prediction(boxes, example)
        if self.training:
            td = self.roi_head.assign_targets(example)
            example["rois"], example["roi_labels"], example["roi_scores"] = td["rois"], td["roi_labels"], td["roi_scores"]
            example["roi_targets_dict"] = td
        for module in self.second_stage:
            example = module(example)
        if self.point_head is not None:
            example = self.point_head(example)
        batch_dict = self.roi_head(example, training=return_loss)
        if not return_loss:
            return self.post_process(batch_dict)
        roi_loss, tb = self.roi_head.get_loss()
        point_loss = 0
        if self.point_head is not None:
            point_loss, tb = self.point_head.get_loss(tb)
        return self.combine_loss(one_stage_loss, point_loss, roi_loss, tb)

    @torch.no_grad()
    def forward_raw(self, points):
        """Inference without a host wait anywhere (what a HIP graph can hold): first stage on the ranked head,
        RoIs padded on the device (`reorder_first_stage_prediction` with the counts left in device memory),
        second stage, scores.  -> (boxes [B, N, L], scores [B, N], labels [B, N], keep [B, N] bool, counts + status
        word of the first stage [B + 1]) with N = NMS_POST_MAXSIZE, or None when the first stage cannot take the
        ranked head.  `keep` marks the rows forward(example, return_loss=False) returns, in its order."""
        sd = self.single_det
        B = len(points)
        bev, feats = sd.extract_feat(dict(points=points, batch_size=B))
        plan = sd.bbox_head.lazy_plan(sd.test_cfg) if bev[0].dtype == torch.bfloat16 else None
        raw = sd.bbox_head.predict_lazy(None, bev, sd.test_cfg, plan, raw=True) if plan is not None else None
        if raw is None:
            return None
        boxes, scores, labels, num = raw
        N = sd.NMS_POST_MAXSIZE
        assert boxes.shape[1] == N, (boxes.shape, N)
        valid = torch.arange(N, device=boxes.device).unsqueeze(0) < num[:B].unsqueeze(1)
        if self.roi_head.code_size == 9:
            boxes = boxes[:, :, [0, 1, 2, 3, 4, 5, 8, 6, 7]]
        example = dict(batch_size=B, metadata=[None] * B, has_class_labels=True, bev_feature=bev[-1], backbone_features=feats,
                       rois=torch.where(valid.unsqueeze(-1), boxes, boxes.new_zeros(())),
                       roi_labels=torch.where(valid, labels + 1, labels.new_zeros(())),
                       roi_scores=torch.where(valid, scores, scores.new_zeros(())))
        for module in self.second_stage:
            example = module(example)
        if self.point_head is not None:
            example = self.point_head(example)
        bd = self.roi_head(example, training=False)
        box, lab = bd["batch_box_preds"], bd["roi_labels"]
        if box.shape[-1] == 9:
            box = box[:, :, [0, 1, 2, 3, 4, 5, 7, 8, 6]]
        score = torch.sqrt(torch.sigmoid(bd["batch_cls_preds"]).view(B, N) * bd["roi_scores"].view(B, N))
        keep = (lab != 0).view(B, N) & (box[..., 3:6] > 0).all(-1)
        return box.contiguous(), score, lab.view(B, N) - 1, keep, num

    def post_process(self, batch_dict):
        """pillar_rcnn.py:133-166: score = sqrt(sigmoid(cls) * roi_score); no second NMS."""
        out = []
        for i in range(batch_dict["batch_size"]):
            box, cls, lab = batch_dict["batch_box_preds"][i], batch_dict["batch_cls_preds"][i], batch_dict["roi_labels"][i]
            if box.shape[-1] == 9:
                box = box[:, [0, 1, 2, 3, 4, 5, 7, 8, 6]]
            score = torch.sqrt(torch.sigmoid(cls).view(-1) * batch_dict["roi_scores"][i].view(-1))
            keep = ((lab != 0).view(-1) & (box[:, 3:6] > 0).all(1)).nonzero().view(-1)   # one host sync per sample
            out.append({"box3d_lidar": box.index_select(0, keep), "scores": score.index_select(0, keep),
                        "label_preds": lab.index_select(0, keep) - 1, "metadata": batch_dict["metadata"][i]})
        return out


class PendingRefined:
    """Two-stage detections of a frame whose kernels are queued but not waited for (the two-stage counterpart of
    bbox_heads.PendingDetections): result() waits for this frame's first-stage status word, then picks the kept rows."""

    def __init__(self, example, tensors, redo):
        self.example, self.redo = example, redo
        self.boxes, self.scores, self.labels, self.keep, num = tensors
        self.host = torch.empty(num.shape, dtype=num.dtype, pin_memory=True)
        self.host.copy_(num, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self.out = None

    def result(self):
        if self.out is not None:
            return self.out
        self.event.synchronize()
        B = self.boxes.shape[0]
        if int(self.host[B]) & 2:            # the first stage's range mask removed ranked cells of an over-full set
            self.out, self.redo = self.redo(), None
            return self.out
        metas = self.example.get("metadata", None) or [None] * B
        self.out = []
        for b in range(B):
            idx = self.keep[b].nonzero().view(-1)
            self.out.append({"box3d_lidar": self.boxes[b].index_select(0, idx), "scores": self.scores[b].index_select(0, idx),
                             "label_preds": self.labels[b].index_select(0, idx), "metadata": metas[b]})
        self.redo = None
        return self.out
