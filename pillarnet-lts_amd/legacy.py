"""Adapters for the 30 legacy config files (configs/nusc/**, configs/waymo/**) that the reference inherits
from CenterPoint and an older PillarNet.  They name types the reference itself does not register any more
(`Voxelization`, `PillarFeatureNet`, `PointPillarsScatter`, `SpMiddlePillarEncoder18`, `SpMiddleResNetFHD`,
`VoxelFeatureExtractorV3`, `RPNG2`, `CenterStrideHead`, `Identity`, ...; det3d's own `build_from_cfg` raises
KeyError on them, det3d/utils/registry.py:63-67) and pass kwargs its live classes no longer take
(`CenterHead(in_channels=256, dataset=..., weight=...)`, `RPNV1(ds_layer_strides=...)`).  With these adapters
every file parses (`Config.fromfile`) and `build_detector(cfg.model, ...)` constructs a model.

No implementation of these types exists in the reference, so behaviour is taken from the config kwargs and
the upstream CenterPoint conventions and is NUMERICALLY UNPINNED:
  * the legacy PillarNet family (Identity reader + SpMiddlePillarEncoder* + RPN*/RPNG2 + CenterHead /
    CenterStrideHead) maps onto the live MI355X path: DynamicPFE at the `pool1` pillar size + PillarResNet18/34
    + the live neck of the nearest shape + CenterHead -- it runs end to end on the HIP kernels;
  * PointPillars (PillarFeatureNet + PointPillarsScatter + RPN + CenterHead) is implemented as upstream
    CenterPoint defines it (dense pillars from the hard voxelizer), in torch;
  * VoxelNet's 3-D sparse backbone (SpMiddleResNetFHD) constructs but raises on forward: 3-D sparse
    convolution is outside this package's 2-D pillar hot path;
  * TwoStageDetector / BEVFeatureExtractor / RoIHead follow det3d/models/detectors/two_stage.py:10-195,
    second_stage/bird_eye_view.py:10-41 and roi_heads/roi_head.py:14-111 (inference and the training branch:
    RoI targets and losses are roi_head_template.py's).
Pipeline step `Voxelization` = the hard voxelizer (det3d/ops/point_cloud/point_cloud_ops.py:112-184), here as a
host-side C routine in the library (`pn_points_to_voxel`); legacy `voxel_generator` dicts
(configs/nusc/voxelnet/nusc_centerpoint_pillarnet_h.py:151-158).
"""
import logging

import numpy as np
import torch
import torch.nn as nn

from . import builder
from .bbox_heads import CenterHead, resolve_legacy_geometry  # noqa: F401
from .detectors import BaseDetector, PillarNet, SingleStageDetector
from .second_stage import RoIHeadTemplate
from .registry import (BACKBONES, DETECTORS, HEADS, NECKS, PIPELINES, READERS, ROI_HEAD, SECOND_STAGE)

_log = logging.getLogger("legacy")


# ---- readers ----------------------------------------------------------------------------------------
@READERS.register_module
class Identity(nn.Module):
    """`reader=dict(type="Identity", pc_range=..., num_input_features=2)`: the points go to the backbone as
    they are (the legacy SpMiddlePillarEncoder* pillarised them itself)."""

    def __init__(self, pc_range=None, num_input_features=2, **kwargs):
        super().__init__()
        self.pc_range, self.num_input_features = pc_range, num_input_features

    def forward(self, data, *args, **kwargs):
        return data


class _PFNLayer(nn.Module):
    """CenterPoint's PFNLayer: Linear(no bias) + BatchNorm1d(eps 1e-3, momentum 0.01) + ReLU, max over the
    points of a pillar; all but the last layer concatenate the max back onto every point."""

    def __init__(self, cin, cout, last):
        super().__init__()
        self.last = last
        self.units = cout if last else cout // 2
        self.linear = nn.Linear(cin, self.units, bias=False)
        self.norm = nn.BatchNorm1d(self.units, eps=1e-3, momentum=0.01)

    def forward(self, x):                                    # [P, N, C]
        x = self.linear(x)
        x = torch.relu(self.norm(x.permute(0, 2, 1).contiguous()).permute(0, 2, 1).contiguous())
        x_max = x.max(dim=1, keepdim=True)[0]
        if self.last:
            return x_max
        return torch.cat([x, x_max.expand(-1, x.shape[1], -1)], dim=2)


@READERS.register_module
class PillarFeatureNet(nn.Module):
    """Hard-voxel pillar feature net (CenterPoint / PointPillars): per point (x, y, z, ..., offset to the
    pillar's mean (3), offset to the pillar's centre (2) [, distance]); PFN layers; max over the pillar."""

    def __init__(self, num_input_features=4, num_filters=(64,), with_distance=False, voxel_size=(0.2, 0.2, 4),
                 pc_range=(0, -40, -3, 70.4, 40, 1), norm_cfg=None, virtual=False, **kwargs):
        super().__init__()
        assert len(num_filters) > 0
        cin = num_input_features + 5 + int(with_distance)
        self.with_distance = with_distance
        widths = [cin] + list(num_filters)
        self.pfn_layers = nn.ModuleList(_PFNLayer(widths[i], widths[i + 1], i == len(widths) - 2)
                                        for i in range(len(widths) - 1))
        self.vx, self.vy = voxel_size[0], voxel_size[1]
        self.x_offset, self.y_offset = self.vx / 2 + pc_range[0], self.vy / 2 + pc_range[1]

    def forward(self, features, num_voxels, coors):           # [P, N, F], [P], [P, 4] (b, z, y, x)
        n = num_voxels.type_as(features).view(-1, 1, 1).clamp(min=1)
        mean = features[:, :, :3].sum(dim=1, keepdim=True) / n
        f_cluster = features[:, :, :3] - mean
        f_center = torch.zeros_like(features[:, :, :2])
        f_center[:, :, 0] = features[:, :, 0] - (coors[:, 3].type_as(features).unsqueeze(1) * self.vx + self.x_offset)
        f_center[:, :, 1] = features[:, :, 1] - (coors[:, 2].type_as(features).unsqueeze(1) * self.vy + self.y_offset)
        parts = [features, f_cluster, f_center]
        if self.with_distance:
            parts.append(torch.norm(features[:, :, :3], 2, 2, keepdim=True))
        x = torch.cat(parts, dim=-1)
        mask = (torch.arange(x.shape[1], device=x.device).view(1, -1) < num_voxels.view(-1, 1)).unsqueeze(-1)
        x = x * mask.type_as(x)
        for pfn in self.pfn_layers:
            x = pfn(x)
        return x.squeeze(1)


@READERS.register_module
class VoxelFeatureExtractorV3(nn.Module):
    """mean of the points of a voxel (CenterPoint's VoxelNet reader)."""

    def __init__(self, num_input_features=4, norm_cfg=None, name="VoxelFeatureExtractorV3", **kwargs):
        super().__init__()
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels, coors=None):
        mean = features[:, :, :self.num_input_features].sum(dim=1) / num_voxels.type_as(features).view(-1, 1).clamp(min=1)
        return mean.contiguous()


@READERS.register_module
class SimpleVoxel(VoxelFeatureExtractorV3):
    pass


# ---- backbones --------------------------------------------------------------------------------------
@BACKBONES.register_module
class PointPillarsScatter(nn.Module):
    """pillar features -> dense BEV canvas [B, C, H, W] (zero where empty)."""

    def __init__(self, num_input_features=64, norm_cfg=None, name="PointPillarsScatter", ds_factor=1, **kwargs):
        super().__init__()
        self.nchannels = num_input_features

    def forward(self, voxel_features, coords, batch_size, input_shape):
        nx, ny = int(input_shape[0]), int(input_shape[1])
        C = voxel_features.shape[1]
        canvas = voxel_features.new_zeros((batch_size, C, ny * nx))
        idx = coords[:, 2].long() * nx + coords[:, 3].long()
        canvas[coords[:, 0].long(), :, idx] = voxel_features
        return canvas.view(batch_size, C, ny, nx)


class _LegacyPillarEncoder(nn.Module):
    """SpMiddlePillarEncoder*(ds_factor, double, pc_range, pillar_cfg=dict(pool0.., pool1=dict(bev=size), ..)):
    pillarises the raw points itself and runs the sparse 2-D residual stack.  Here: DynamicPFE at the pool1
    pillar size + the live PillarResNet of matching depth (dense conv5 tail when a pool4 scale is listed)."""
    DEPTH = "18"

    def __init__(self, ds_factor=8, double=2, pc_range=None, pillar_cfg=None, num_input_features=2,
                 in_channels=5, **kwargs):
        super().__init__()
        from .backbones import PillarResNet18, PillarResNet18S, PillarResNet34, PillarResNet34S
        from .readers import DynamicPFE
        pillar_cfg = pillar_cfg or {}
        size = float(pillar_cfg.get("pool1", {}).get("bev", 0.075))
        self.ds_factor = ds_factor
        self.pfe = DynamicPFE(in_channels=in_channels, num_filters=(32,), pillar_size=size, pc_range=list(pc_range))
        dense_tail = "pool4" in pillar_cfg
        cls = {("18", True): PillarResNet18, ("18", False): PillarResNet18S,
               ("34", True): PillarResNet34, ("34", False): PillarResNet34S}[(self.DEPTH, dense_tail)]
        self.net = cls(in_channels=32)
        self.backbone_channels, self.backbone_strides = self.net.backbone_channels, self.net.backbone_strides

    def forward(self, data):
        sp = self.pfe(data)
        dtype = getattr(self, "compute_dtype", sp.features_cap.dtype)
        if sp.features_cap.dtype != dtype:
            sp = sp.replace_feature(sp.features_cap.to(dtype))
        return self.net(sp)


@BACKBONES.register_module
class SpMiddlePillarEncoder(_LegacyPillarEncoder):
    pass


@BACKBONES.register_module
class SpMiddlePillarEncoder18(_LegacyPillarEncoder):
    pass


@BACKBONES.register_module
class SpMiddlePillarEncoderCSP18(_LegacyPillarEncoder):
    pass


@BACKBONES.register_module
class SpMiddlePillarEncoder34(_LegacyPillarEncoder):
    DEPTH = "34"


@BACKBONES.register_module
class SpMiddleResNetFHD(nn.Module):
    """CenterPoint's 3-D sparse-conv voxel backbone.  Constructible (so VoxelNet configs load); forward needs
    3-D sparse convolution, which this package -- a 2-D pillar path -- does not provide."""

    def __init__(self, num_input_features=128, norm_cfg=None, name="SpMiddleResNetFHD", ds_factor=8, **kwargs):
        super().__init__()
        self.num_input_features, self.ds_factor = num_input_features, ds_factor

    def forward(self, *args, **kwargs):
        raise NotImplementedError("SpMiddleResNetFHD: 3-D sparse convolution is not part of the pillar hot path")


# ---- necks ---------------------------------------------------------------------------------------------
def _legacy_neck_kwargs(kw):
    """ds_* / us_* / num_input_features of the legacy necks -> (layer_nums, in_channels, num_filters)."""
    nin = kw.pop("num_input_features")
    nin = list(nin) if isinstance(nin, (list, tuple)) else [nin]
    ds = list(kw.pop("ds_num_filters"))
    us = list(kw.pop("us_num_filters", ds))
    for k in ("ds_layer_strides", "us_layer_strides", "use_csp", "name"):
        kw.pop(k, None)
    return nin, ds, us


@NECKS.register_module
class RPNG2(nn.Module):
    """two-scale FPN neck of the legacy PillarNet configs -> the live RPNG (det3d/models/necks/rpn.py:276-355)."""

    def __new__(cls, layer_nums, **kw):
        from .necks import RPNG
        nin, ds, us = _legacy_neck_kwargs(kw)
        # both output maps get the width of the last up-sampling entry: the legacy stride heads take one
        # `in_channels` for every stride
        return RPNG(layer_nums=layer_nums, in_channels=[nin[0], nin[-1], nin[-1] // 2],
                    num_filters=[us[-1], us[-1]], logger=kw.get("logger"))


def legacy_rpn_v(cls_name, layer_nums, kw):
    """RPNV1 / RPNV2 given legacy kwargs (ds_num_filters, num_input_features, ...)."""
    from . import necks
    nin, ds, _ = _legacy_neck_kwargs(kw)
    live = getattr(necks, cls_name)
    if len(nin) == 1:
        nin = nin * 2
    return live(layer_nums=layer_nums, num_filters=ds[0], in_channels=nin, logger=kw.get("logger"))


# ---- heads -----------------------------------------------------------------------------------------------
def legacy_head_kwargs(kw):
    """CenterHead(in_channels=int, dataset, weight, share_conv_channel, dcn_head, iou_reg, order_class_names,
    init_bias, num_hm_conv) of the legacy configs -> the live constructor's kwargs.  Task strides, pillar size
    and range are not in the legacy head dict: the detector fills them from test_cfg (`out_size_factor`,
    `voxel_size`, `pc_range`)."""
    kw = dict(kw)
    if isinstance(kw.get("in_channels"), int):
        n_strides = len({t.get("stride", 0) for t in kw["tasks"]})
        kw["in_channels"] = [kw["in_channels"]] * n_strides
    if "share_conv_channel" in kw:
        kw["share_channel"] = kw.pop("share_conv_channel")
    if "iou_reg" in kw:
        kw["reg_iou"] = kw.pop("iou_reg")
    if kw.pop("dcn_head", False):
        _log.warning("legacy CenterHead: dcn_head=True is not built (deformable conv); plain heads are used")
    for k in ("dataset", "weight", "order_class_names", "init_bias", "num_hm_conv", "name"):
        kw.pop(k, None)
    tasks = []
    for t in kw["tasks"]:
        t = dict(t)
        t.setdefault("stride", 0)                      # 0: resolved by the detector from test_cfg
        tasks.append(t)
    kw["tasks"] = tasks
    return kw


@HEADS.register_module
class CenterStrideHead(CenterHead):
    def __init__(self, **kw):
        super().__init__(**legacy_head_kwargs(kw))


@HEADS.register_module
class RPNHead(CenterStrideHead):
    pass


# ---- detectors ---------------------------------------------------------------------------------------------
@DETECTORS.register_module
class PointPillars(SingleStageDetector):
    """reader(voxels, num_points, coordinates) -> scatter -> neck -> head (CenterPoint's point_pillars.py)."""

    def extract_feat(self, data):
        feats = self.reader(data["features"], data["num_voxels"], data["coors"])
        x = self.backbone(feats, data["coors"], data["batch_size"], data["input_shape"])
        return self.neck(x) if self.with_neck else x

    def forward(self, example, return_loss=True, **kwargs):
        data = dict(features=example["voxels"], num_voxels=example["num_points"], coors=example["coordinates"],
                    batch_size=len(example["num_voxels"]), input_shape=example["shape"][0])
        x = self.extract_feat(data)
        if isinstance(x, tuple) and x[0].dim() == 3:       # RPN hands back the batch split into samples
            x = torch.stack(x)
        x = x if isinstance(x, (list, tuple)) else (x,)
        preds = self.bbox_head(x)
        if return_loss:
            return self.bbox_head.loss(example, preds, self.train_cfg)
        return self.bbox_head.predict(example, preds, self.test_cfg)


    def forward_two_stage(self, example, return_loss=True, **kwargs):
        """upstream CenterPoint point_pillars.py forward_two_stage: (boxes, BEV feature map, loss)."""
        data = dict(features=example["voxels"], num_voxels=example["num_points"], coors=example["coordinates"],
                    batch_size=len(example["num_voxels"]), input_shape=example["shape"][0])
        x = self.extract_feat(data)
        if isinstance(x, tuple) and x[0].dim() == 3:
            x = torch.stack(x)
        bev = x[0] if isinstance(x, (list, tuple)) else x
        preds = self.bbox_head((bev,))
        detached = [{k: v.detach() for k, v in p.items()} for p in preds]
        boxes = self.bbox_head.predict(example, detached, self.test_cfg)
        loss = self.bbox_head.loss(example, preds, self.train_cfg) if return_loss else None
        return boxes, bev, loss


@DETECTORS.register_module
class VoxelNet(SingleStageDetector):
    """det3d/models/detectors/voxelnet.py:6-57 (its backbone here raises on forward, see SpMiddleResNetFHD)."""

    def extract_feat(self, data):
        x, voxel_feature = self.backbone(self.reader(data["features"], data["num_voxels"]), data["coors"],
                                         data["batch_size"], data["input_shape"])
        return (self.neck(x) if self.with_neck else x), voxel_feature

    def forward(self, example, return_loss=True, **kwargs):
        data = dict(features=example["voxels"], num_voxels=example["num_points"], coors=example["coordinates"],
                    batch_size=len(example["num_voxels"]), input_shape=example["shape"][0])
        x, _ = self.extract_feat(data)
        preds = self.bbox_head(x if isinstance(x, (list, tuple)) else (x,))
        if return_loss:
            return self.bbox_head.loss(example, preds, self.train_cfg)
        return self.bbox_head.predict(example, preds, self.test_cfg)


def bilinear_interpolate_torch(im, x, y):
    """det3d/core/utils/center_utils.py:91-120: im [H, W, C], x / y [N] in cell units; corners clamped to the map, weights
    from the clamped corners (so points beyond the edge fade out exactly as in the reference)."""
    x0, y0 = torch.floor(x).long(), torch.floor(y).long()
    x1, y1 = x0 + 1, y0 + 1
    x0, x1 = torch.clamp(x0, 0, im.shape[1] - 1), torch.clamp(x1, 0, im.shape[1] - 1)
    y0, y1 = torch.clamp(y0, 0, im.shape[0] - 1), torch.clamp(y1, 0, im.shape[0] - 1)
    wa = (x1.type_as(x) - x) * (y1.type_as(y) - y)
    wb = (x1.type_as(x) - x) * (y - y0.type_as(y))
    wc = (x - x0.type_as(x)) * (y1.type_as(y) - y)
    wd = (x - x0.type_as(x)) * (y - y0.type_as(y))
    return (im[y0, x0] * wa.unsqueeze(1) + im[y1, x0] * wb.unsqueeze(1) + im[y0, x1] * wc.unsqueeze(1) +
            im[y1, x1] * wd.unsqueeze(1))


@SECOND_STAGE.register_module
class BEVFeatureExtractor(nn.Module):
    """det3d/models/second_stage/bird_eye_view.py:10-41: bilinear BEV features at the box centre points."""

    def __init__(self, pc_start, voxel_size, out_stride):
        super().__init__()
        self.pc_start, self.voxel_size, self.out_stride = pc_start, voxel_size, out_stride

    def absl_to_relative(self, absolute):
        a1 = (absolute[..., 0] - self.pc_start[0]) / (self.voxel_size[0] * self.out_stride)
        a2 = (absolute[..., 1] - self.pc_start[1]) / (self.voxel_size[0] * self.out_stride)
        return a1, a2

    def forward(self, example, batch_centers, num_point):
        out = []
        for b in range(len(example["bev_feature"])):
            xs, ys = self.absl_to_relative(batch_centers[b])
            fmap = bilinear_interpolate_torch(example["bev_feature"][b], xs, ys)
            if num_point > 1:
                sec = len(fmap) // num_point
                fmap = torch.cat([fmap[i * sec:(i + 1) * sec] for i in range(num_point)], dim=1)
            out.append(fmap)
        return out


@SECOND_STAGE.register_module
class BEVFeatureNeck(nn.Module):
    """second-stage map fusion of the legacy PillarRCNN config (feature_sources 'x_conv3', ...; extra_scale)
    -> the live BEVStrideFeature (det3d/models/second_stage/bev_interpolation.py:162-308)."""

    def __new__(cls, feature_sources, extra_scale=1, in_channels=256, bev_channels=None, **kw):
        from .second_stage import BEVStrideFeature
        return BEVStrideFeature(feature_sources=[s[2:] if s.startswith("x_") else s for s in feature_sources],
                                in_channels=bev_channels or in_channels, **kw)


@SECOND_STAGE.register_module
class BEVFeatureStrideNeck(BEVFeatureNeck):
    """the same with `bev_channels` naming the width of the neck map it reads."""


@ROI_HEAD.register_module
class RoIHead(RoIHeadTemplate):
    """det3d/models/roi_heads/roi_head.py:14-111: shared Conv1d(k1)+BN+ReLU stack, then class and box-residual
    branches (same parameter names: shared_fc_layer / cls_layers / reg_layers); targets and losses are the
    template's (roi_head_template.py)."""

    def __init__(self, input_channels, model_cfg, num_class=1, code_size=7, add_box_param=False, test_cfg=None):
        super().__init__(num_class=num_class, model_cfg=model_cfg)
        self.test_cfg, self.code_size, self.add_box_param = test_cfg, code_size, add_box_param
        pre, layers = input_channels, []
        shared = list(model_cfg["SHARED_FC"])
        for k, c in enumerate(shared):
            layers += [nn.Conv1d(pre, c, kernel_size=1, bias=False), nn.BatchNorm1d(c), nn.ReLU()]
            pre = c
            if k != len(shared) - 1 and model_cfg["DP_RATIO"] > 0:
                layers.append(nn.Dropout(model_cfg["DP_RATIO"]))
        self.shared_fc_layer = nn.Sequential(*layers)
        self.cls_layers = self.make_fc_layers(pre, num_class, model_cfg["CLS_FC"])
        self.reg_layers = self.make_fc_layers(pre, code_size, model_cfg["REG_FC"])
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.reg_layers[-1].weight, mean=0, std=0.001)

    def forward(self, batch_dict, training=True):
        """roi_head.py:70-111."""
        batch_dict["batch_size"] = len(batch_dict["rois"])
        if training:
            td = batch_dict.get("roi_targets_dict", None)
            if td is None:
                td = self.assign_targets(batch_dict)
                batch_dict["rois"], batch_dict["roi_labels"], batch_dict["roi_scores"] = \
                    td["rois"], td["roi_labels"], td["roi_scores"]
        if self.add_box_param:
            batch_dict["roi_features"] = torch.cat([batch_dict["roi_features"], batch_dict["rois"],
                                                    batch_dict["roi_scores"].unsqueeze(-1)], dim=-1)
        feats = batch_dict["roi_features"].reshape(-1, 1, batch_dict["roi_features"].shape[-1]).permute(0, 2, 1).contiguous()
        shared = self.shared_fc_layer(feats)
        rcnn_cls = self.cls_layers(shared).transpose(1, 2).contiguous().squeeze(dim=1)
        rcnn_reg = self.reg_layers(shared).transpose(1, 2).contiguous().squeeze(dim=1)
        if training:
            td["rcnn_cls"], td["rcnn_reg"] = rcnn_cls, rcnn_reg
            self.forward_ret_dict = td
            return batch_dict
        cls, boxes = self.generate_predicted_boxes(batch_dict["batch_size"], batch_dict["rois"], rcnn_cls, rcnn_reg)
        batch_dict["batch_cls_preds"], batch_dict["batch_box_preds"] = cls, boxes
        batch_dict["cls_preds_normalized"] = False
        return batch_dict


@DETECTORS.register_module
class TwoStageDetector(BaseDetector):
    """det3d/models/detectors/two_stage.py:10-195 (the legacy first stages it wraps are the adapters above; their
    numerics are unpinned, see the module docstring)."""

    def __init__(self, first_stage_cfg, second_stage_modules, roi_head, point_head=None, num_point=1, freeze=False,
                 use_final_feature=False, NMS_POST_MAXSIZE=None, **kwargs):
        super().__init__()
        self.NMS_POST_MAXSIZE = NMS_POST_MAXSIZE
        self.single_det = builder.build_detector(first_stage_cfg, **kwargs)
        if freeze:
            for p in self.single_det.parameters():
                p.requires_grad = False
        self.bbox_head = self.single_det.bbox_head
        self.test_cfg = self.single_det.test_cfg
        self.second_stage = nn.ModuleList(builder.build(m, SECOND_STAGE) for m in second_stage_modules)
        if point_head is not None:
            from .registry import POINT_HEAD
            self.point_head = builder.build(point_head, POINT_HEAD)
        self.roi_head = builder.build(roi_head, ROI_HEAD)
        self.num_point, self.use_final_feature = num_point, use_final_feature

    def get_box_center(self, boxes):
        """two_stage.py:52-79: the box centre, or centre + the four face centres of the BEV box (num_point 5)."""
        centers = []
        for box in boxes:
            b = box["box3d_lidar"]
            if self.num_point == 1 or len(b) == 0:
                centers.append(b[:, :3])
            elif self.num_point == 5:
                c2, h, d2, ry = b[:, :2], b[:, 2:3], b[:, 3:5], b[:, -1]
                # box_torch_ops.center_to_corner_box2d (det3d/core/bbox/box_torch_ops.py:198-217): corners clockwise
                # from the minimum point, rotated clockwise by ry
                norm = b.new_tensor([[-0.5, -0.5], [-0.5, 0.5], [0.5, 0.5], [0.5, -0.5]])
                corners = d2.view(-1, 1, 2) * norm.view(1, 4, 2)
                sin, cos = torch.sin(ry).view(-1, 1), torch.cos(ry).view(-1, 1)
                corners = torch.stack([corners[..., 0] * cos + corners[..., 1] * sin,
                                       -corners[..., 0] * sin + corners[..., 1] * cos], dim=-1) + c2.view(-1, 1, 2)
                mid = lambda i, j: torch.cat([(corners[:, i] + corners[:, j]) / 2, h], dim=-1)
                centers.append(torch.cat([b[:, :3], mid(0, 1), mid(2, 3), mid(0, 3), mid(1, 2)], dim=0))
            else:
                raise NotImplementedError()
        return centers

    def reorder_first_stage_pred_and_feature(self, first_pred, example, features):
        """two_stage.py:81-118: pad every sample to NMS_POST_MAXSIZE RoIs, labels shifted by +1."""
        B, L, N = len(first_pred), first_pred[0]["box3d_lidar"].shape[1], self.NMS_POST_MAXSIZE
        F = sum(feat[0].shape[-1] for feat in features)
        ref = first_pred[0]["box3d_lidar"]
        rois, scores = ref.new_zeros((B, N, L)), first_pred[0]["scores"].new_zeros((B, N))
        labels = first_pred[0]["label_preds"].new_zeros((B, N), dtype=torch.long)
        feats = features[0][0].new_zeros((B, N, F))
        for i in range(B):
            n = features[0][i].shape[0]
            box = first_pred[i]["box3d_lidar"]
            if self.roi_head.code_size == 9:
                box = box[:, [0, 1, 2, 3, 4, 5, 8, 6, 7]]
            rois[i, :n], labels[i, :n], scores[i, :n] = box, first_pred[i]["label_preds"] + 1, first_pred[i]["scores"]
            feats[i, :n] = torch.cat([feat[i] for feat in features], dim=-1)
        example.update(rois=rois, roi_labels=labels, roi_scores=scores, roi_features=feats, has_class_labels=True)
        return example

    def post_process(self, batch_dict):
        """two_stage.py:120-150: score = sqrt(sigmoid(cls) * first-stage score), padded RoIs dropped, no second NMS."""
        out = []
        for i in range(batch_dict["batch_size"]):
            box, cls, lab = batch_dict["batch_box_preds"][i], batch_dict["batch_cls_preds"][i], batch_dict["roi_labels"][i]
            if box.shape[-1] == 9:
                box = box[:, [0, 1, 2, 3, 4, 5, 7, 8, 6]]
            score = torch.sqrt(torch.sigmoid(cls).view(-1) * batch_dict["roi_scores"][i].view(-1))
            mask = (lab != 0).view(-1)
            out.append({"box3d_lidar": box[mask], "scores": score[mask], "label_preds": lab[mask] - 1,
                        "metadata": batch_dict["metadata"][i]})
        return out

    def combine_loss(self, one_stage_loss, roi_loss, tb_dict):
        """two_stage.py:43-50."""
        one_stage_loss["loss"][0] = one_stage_loss["loss"][0] + roi_loss
        for _ in range(len(one_stage_loss["loss"])):
            one_stage_loss["roi_reg_loss"].append(tb_dict["rcnn_loss_reg"])
            one_stage_loss["roi_cls_loss"].append(tb_dict["rcnn_loss_cls"])
        return one_stage_loss

    def forward(self, example, return_loss=True, **kwargs):
        """two_stage.py:152-195."""
        out = self.single_det.forward_two_stage(example, return_loss, **kwargs)
        if len(out) == 5:
            first, bev, example["voxel_feature"], final, one_stage_loss = out
        elif len(out) == 3:
            first, bev, one_stage_loss = out
            final = None
        else:
            raise NotImplementedError
        src = final if self.use_final_feature else bev
        example["bev_feature"] = src.permute(0, 2, 3, 1).contiguous()
        centers = self.get_box_center(first)
        if self.roi_head.code_size == 7 and return_loss is True:
            example["gt_boxes_and_cls"] = example["gt_boxes_and_cls"][:, :, [0, 1, 2, 3, 4, 5, 6, -1]]   # drop velocity
        features = [m.forward(example, centers, self.num_point) for m in self.second_stage]
        example = self.reorder_first_stage_pred_and_feature(first, example, features)
        batch_dict = self.roi_head(example, training=return_loss)
        if return_loss:
            roi_loss, tb = self.roi_head.get_loss()
            return self.combine_loss(one_stage_loss, roi_loss, tb)
        return self.post_process(batch_dict)


# ---- pipeline: the hard voxelizer -----------------------------------------------------------------------------
def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000):
    """det3d/ops/point_cloud/point_cloud_ops.py:112-184 on the host (the reference runs it in dataloader
    workers under numba): first-come voxel order, `max_points` per voxel, at most `max_voxels` voxels.
    -> (voxels [M, max_points, F], coordinates [M, 3] int32 (z, y, x when reverse_index), num_points [M])."""
    import ctypes
    from . import _lib
    points = np.ascontiguousarray(points, dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    rng = np.asarray(coors_range, dtype=np.float32)
    grid = np.round((rng[3:] - rng[:3]) / vs).astype(np.int32)
    shape = tuple(int(v) for v in (grid[::-1] if reverse_index else grid))
    num = np.zeros(max_voxels, np.int32)
    cmap = np.full(shape, -1, np.int32)
    voxels = np.zeros((max_voxels, max_points, points.shape[1]), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    fn = _lib.lib().pn_points_to_voxel
    m = fn(points.ctypes.data_as(ctypes.c_void_p), points.shape[0], points.shape[1], vs.ctypes.data_as(ctypes.c_void_p),
           rng.ctypes.data_as(ctypes.c_void_p), int(reverse_index), int(max_points), int(max_voxels),
           cmap.ctypes.data_as(ctypes.c_void_p), voxels.ctypes.data_as(ctypes.c_void_p),
           coors.ctypes.data_as(ctypes.c_void_p), num.ctypes.data_as(ctypes.c_void_p))
    if m < 0:
        _lib.check(m, "pn_points_to_voxel")
    return voxels[:m], coors[:m], num[:m]


@PIPELINES.register_module
class Voxelization:
    """det3d's `Voxelization(cfg=voxel_generator)` pipeline step of the legacy configs: range, voxel_size,
    max_points_in_voxel, max_voxel_num ([train, test] or int)."""

    def __init__(self, **kwargs):
        cfg = kwargs.get("cfg", kwargs)
        self.range, self.voxel_size = cfg["range"], cfg["voxel_size"]
        self.max_points_in_voxel = cfg["max_points_in_voxel"]
        mv = cfg["max_voxel_num"]
        self.max_voxel_num = list(mv) if isinstance(mv, (list, tuple)) else [mv, mv]
        self.double_flip = cfg.get("double_flip", False)
        vs, rng = np.asarray(self.voxel_size, np.float32), np.asarray(self.range, np.float32)
        self.grid_size = np.round((rng[3:] - rng[:3]) / vs).astype(np.int64)

    def __call__(self, res, info):
        train = res.get("mode", "val") == "train"
        max_voxels = self.max_voxel_num[0 if train else 1]
        voxels, coords, num = points_to_voxel(res["lidar"]["points"], self.voxel_size, self.range,
                                              self.max_points_in_voxel, True, max_voxels)
        res["lidar"]["voxels"] = dict(voxels=voxels, coordinates=coords, num_points=num,
                                      num_voxels=np.array([voxels.shape[0]], dtype=np.int64), shape=self.grid_size,
                                      range=self.range, size=self.voxel_size)
        return res, info


@PIPELINES.register_module
class Empty:
    def __init__(self, **kwargs):
        pass

    def __call__(self, res, info):
        return res, info
