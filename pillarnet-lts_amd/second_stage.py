"""Pillar R-CNN second stage (inference): SECOND_STAGE.BEVStrideFeature, ROI_HEAD.RoIMIXHead,
POINT_HEAD.PointHead, DETECTORS.PillarRCNN -- det3d/models/second_stage/bev_interpolation.py:162-308,
det3d/models/roi_heads/{roi_mix_head,roi_head_template,mlp_layers}.py,
det3d/models/point_heads/point_head_simple.py, det3d/models/detectors/pillar_rcnn.py.

Module trees / parameter names are the reference's.  The RoI pillar pooling (7x7 rotated grid +
bilinear interpolation) is one HIP launch (`pn_roi_grid_pool`); the small 1-D conv MLPs of the RoI
head stay torch ops.  Training of the second stage (ProposalTargetLayer, RoI / point losses) is not
built: `forward(return_loss=True)` raises."""
import numpy as np
import torch
import torch.nn as nn

from . import _lib, builder, dense_hip
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
            if s < 1:
                raise NotImplementedError("sparse lateral convs (kernel = stride > 1) are not built on the HIP path")
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
        lat = [dense_hip.compile_stack(m) for m in self.lat_conv]
        return td, lat, dense_hip.compile_stack(self.fusion_conv)

    def fused_rows_hip(self, example):
        """bf16 on the gather-GEMM: the up/lateral maps land in channel slices of one buffer (no concat),
        the 3x3 fusion conv + bias + BN + ReLU is one launch.  -> Rows [B*H*W, share_channels] bf16."""
        td, lat, fusion = dense_hip.PLANS.get(self, self._compile)
        x = dense_hip.Rows.from_nchw(example["bev_feature"])
        widths = [td[-1].cout] + [l[-1].cout for l in lat]
        up = 2 ** sum(isinstance(s, dense_hip.UpStep) for s in td)
        B, H, W = x.B, x.H * up, x.W * up
        cat = torch.empty((B * H * W, sum(widths)), dtype=torch.bfloat16, device=x.t.device)
        dense_hip.run_stack(td, x, cat[:, :widths[0]])
        c0 = widths[0]
        for k, src in enumerate(self.lat_conv_name):
            f = example["backbone_features"][src]
            if isinstance(f, SparseConvTensor):
                f = f.dense()
            dense_hip.run_stack(lat[k], dense_hip.Rows.from_nchw(f.to(torch.bfloat16)), cat[:, c0:c0 + widths[k + 1]])
            c0 += widths[k + 1]
        return dense_hip.run_stack(fusion, dense_hip.Rows(cat, B, H, W))

    def fused_map(self, example):
        if getattr(self, "_dense_hip", False) and example["bev_feature"].dtype == torch.bfloat16:
            return self.fused_rows_hip(example).nchw()
        ref = next(self.parameters())
        cast = lambda t: t if t.dtype == ref.dtype else t.to(ref.dtype)
        feats = [self.top_down_conv(cast(example["bev_feature"]))]
        for k, src in enumerate(self.lat_conv_name):
            f = example["backbone_features"][src]
            if isinstance(f, SparseConvTensor):
                f = f.dense()
            feats.append(self.lat_conv[k](cast(f)))
        return self.fusion_conv(torch.cat(feats, dim=1))

    def forward(self, example):
        B, R = example["rois"].shape[:2]
        r = dense_hip.Rows.from_nchw(self.fused_map(example))
        pooled, pts = roi_grid_pool_rows(example["rois"], r.t, r.B, r.H, r.W, self.grid_size,
                                         self.out_stride * self.pillar_size, self.point_cloud_range[0],
                                         self.point_cloud_range[1])
        example["roi_features"] = pooled.view(B, R, -1)
        example["point_features"] = pooled
        example["point_coords"] = pts
        return example


def rotate_points_along_z(points, angle):
    """box_torch_ops.py:120-138."""
    c, s = torch.cos(angle), torch.sin(angle)
    z, o = angle.new_zeros(points.shape[0]), angle.new_ones(points.shape[0])
    rot = torch.stack((c, -s, z, s, c, z, z, z, o), dim=1).view(-1, 3, 3).float()
    return torch.cat((torch.matmul(points[:, :, 0:3], rot), points[:, :, 3:]), dim=-1)


def _fc_stack(cfg_list, cin, cout, dp_ratio, conv=True):
    layers, pre = [], cin
    for k, c in enumerate(cfg_list):
        layers += [nn.Conv1d(pre, c, kernel_size=1, bias=False), build_norm_layer(_BN1, c)[1], nn.ReLU()]
        pre = c
        if dp_ratio >= 0 and k == 0:
            layers.append(nn.Dropout(dp_ratio))
    layers.append(nn.Conv1d(pre, cout, kernel_size=1, bias=True))
    return nn.Sequential(*layers)


@ROI_HEAD.register_module
class RoIMIXHead(nn.Module):
    """roi_mix_head.py:17-122 + roi_head_template.py:22-40,189-219 (inference)."""

    def __init__(self, in_channels, model_cfg, num_class=1, code_size=7, add_box_param=False, test_cfg=None,
                 mixer_type=None, num_patches=49, **kwargs):
        super().__init__()
        self.model_cfg, self.test_cfg, self.num_class = model_cfg, test_cfg, num_class
        self.code_size, self.add_box_param, self.num_patches = code_size, add_box_param, num_patches
        if mixer_type in ("MLPMixer", "ResMLP"):
            raise NotImplementedError(f"mixer_type {mixer_type!r} is not built (the live config uses '')")
        self.mlp_mixer = nn.Sequential()
        pre = in_channels * num_patches
        shared, fcs = [], list(model_cfg["SHARED_FC"])
        for k, c in enumerate(fcs):
            shared += [nn.Conv1d(pre, c, kernel_size=1, bias=False), build_norm_layer(_BN1, c)[1], nn.ReLU()]
            pre = c
            if k != len(fcs) - 1 and model_cfg["DP_RATIO"] > 0:
                shared.append(nn.Dropout(model_cfg["DP_RATIO"]))
        self.shared_fc_layer = nn.Sequential(*shared)
        self.cls_layers = _fc_stack(list(model_cfg["CLS_FC"]), pre, num_class, model_cfg["DP_RATIO"])
        self.reg_layers = _fc_stack(list(model_cfg["REG_FC"]), pre, code_size, model_cfg["DP_RATIO"])
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.reg_layers[-1].weight, mean=0, std=0.001)

    def generate_predicted_boxes(self, batch_size, rois, cls_preds, box_preds):
        code = box_preds.shape[-1]
        cls = cls_preds.view(batch_size, -1, cls_preds.shape[-1])
        ry, xyz = rois[:, :, 6].reshape(-1), rois[:, :, 0:3].reshape(-1, 3)
        local = rois.clone().detach()
        local[:, :, 0:3] = 0
        boxes = (box_preds.view(batch_size, -1, code) + local).view(-1, code)
        boxes = rotate_points_along_z(boxes.unsqueeze(1), ry).squeeze(1)
        boxes[:, 0:3] += xyz
        return cls, boxes.view(batch_size, -1, code)

    def forward(self, batch_dict, training=True):
        if training:
            raise NotImplementedError("RoI head training (ProposalTargetLayer, losses) is not built")
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


@POINT_HEAD.register_module
class PointHead(nn.Module):
    """point_head_simple.py:14-98 (the auxiliary foreground classifier on the RoI grid points)."""

    def __init__(self, in_channels, num_class, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg, self.num_class = model_cfg, num_class
        layers, c_in = [], in_channels
        for c in model_cfg["CLS_FC"]:
            layers += [nn.Linear(c_in, c, bias=False), build_norm_layer(_BN1, c)[1], nn.ReLU()]
            c_in = c
        layers.append(nn.Linear(c_in, 1, bias=True))
        self.cls_layers = nn.Sequential(*layers)

    def forward(self, batch_dict):
        if self.training:
            raise NotImplementedError("point-head target assignment / loss is not built")
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

    def forward(self, example, return_loss=True, **kwargs):
        if return_loss:
            raise NotImplementedError("Pillar R-CNN training is not built (first stage trains through PillarNet)")
        example = dict(example)
        example["batch_size"] = len(example["metadata"])
        boxes, bev, backbone_features, _ = self.single_det.forward_two_stage(example, False, **kwargs)
        example["bev_feature"], example["backbone_features"] = bev[-1], backbone_features
        example = self.reorder_first_stage_prediction(boxes, example)
        for module in self.second_stage:
            example = module(example)
        if self.point_head is not None:
            example = self.point_head(example)
        return self.post_process(self.roi_head(example, training=False))

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
