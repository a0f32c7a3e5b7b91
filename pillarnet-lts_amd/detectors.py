"""DETECTORS registry: `PillarNet`, `SingleStageDetector`, `BaseDetector`
(det3d/models/detectors/{base,single_stage,pillarnet}.py)."""
import torch
import torch.nn as nn

from . import builder
from .center_utils import set_by_task_cfg
from .checkpoint import load_checkpoint
from .registry import DETECTORS


class BaseDetector(nn.Module):
    def __init__(self):
        super().__init__()
        self.fp16_enabled = False

    with_reader = property(lambda self: getattr(self, "reader", None) is not None)
    with_neck = property(lambda self: getattr(self, "neck", None) is not None)
    with_shared_head = property(lambda self: getattr(self, "shared_head", None) is not None)
    with_bbox = property(lambda self: getattr(self, "bbox_head", None) is not None)
    with_mask = property(lambda self: getattr(self, "mask_head", None) is not None)


@DETECTORS.register_module
class SingleStageDetector(BaseDetector):
    def __init__(self, reader, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None):
        super().__init__()
        self.reader = builder.build_reader(reader)
        self.backbone = builder.build_backbone(backbone)
        if neck is not None:
            self.neck = builder.build_neck(neck)
        self.bbox_head = builder.build_head(bbox_head)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        if hasattr(self.bbox_head, "task_strides"):
            from .bbox_heads import resolve_legacy_geometry      # legacy head dicts carry no stride / pillar size
            resolve_legacy_geometry(self.bbox_head, self.test_cfg)
        self.compute_dtype = torch.float32
        self.init_weights(pretrained=pretrained)

    def init_weights(self, pretrained=None):
        if pretrained is None:
            return
        # single_stage.py:31-38 swallows every error here; only the missing-file case is kept soft
        try:
            load_checkpoint(self, pretrained, strict=False)
            print("init weight from {}".format(pretrained))
        except IOError:
            print("no pretrained model at {}".format(pretrained))

    def extract_feat(self, data):
        x = self.backbone(self.reader(data))
        return self.neck(x) if self.with_neck else x

    def forward(self, example, return_loss=True, **kwargs):
        pass

    # -- MI355X inference set-up ----------------------------------------------------------------
    def to_inference(self, dtype=torch.bfloat16, device="cuda", dense_backend="hip"):
        """eval mode for the MI355X inference path.

        dtype bf16: sparse stages run bf16 MFMA with fp32 accumulation.  dense_backend "hip" runs
        conv5 / neck / head through the same gather-GEMM kernel (plans compiled from the modules);
        "miopen" keeps them on PyTorch-ROCm in bf16 channels-last.  dtype fp32 (the parity path)
        always uses the torch modules for the dense part."""
        assert dense_backend in ("hip", "miopen")
        self.eval().to(device)
        self.compute_dtype = dtype
        use_hip = dense_backend == "hip" and dtype == torch.bfloat16
        for m in (self.backbone, getattr(self, "neck", None), self.bbox_head):
            if m is not None:
                m._dense_hip = use_hip
        inner = getattr(self.backbone, "net", None)      # legacy encoders wrap the live sparse stack
        if inner is not None:
            inner._dense_hip = use_hip
        dense = [getattr(inner or self.backbone, "conv5", None), getattr(self, "neck", None), self.bbox_head]
        for m in dense:
            if m is None:
                continue
            if use_hip or dtype == torch.float32:
                m.float()          # plans convert weights themselves and fold BN from fp32 statistics
            else:
                m.to(dtype=dtype, memory_format=torch.channels_last)
        if not use_hip and dtype != torch.float32:
            for th in getattr(self.bbox_head, "task_heads", []):
                th.keep_output_convs_fp32()
        return self


@DETECTORS.register_module
class PillarNet(SingleStageDetector):
    """pillarnet.py:6-82."""

    def __init__(self, reader, backbone, neck, bbox_head, train_cfg=None, test_cfg=None,
                 pretrained=None):
        super().__init__(reader, backbone, neck, bbox_head, train_cfg, test_cfg, pretrained)
        post = self.test_cfg.nms.nms_post_max_size
        self.NMS_POST_MAXSIZE = sum(post) if isinstance(post, list) else post
        self.num_classes = self.bbox_head.num_classes
        self.test_cfg = set_by_task_cfg(self.test_cfg, self.bbox_head.num_classes)

    def extract_feat(self, data):
        sp = self.reader(data)
        if isinstance(sp, dict):     # legacy `Identity` reader: the backbone pillarises the points itself
            self.backbone.compute_dtype = self.compute_dtype
        elif self.compute_dtype != sp.features_cap.dtype:
            sp = sp.replace_feature(sp.features_cap.to(self.compute_dtype))
        pillar_features = self.backbone(sp)
        bev_features = self.neck(pillar_features) if self.with_neck else None
        return bev_features, pillar_features

    def forward(self, example, return_loss=True, **kwargs):
        data = dict(points=example["points"], batch_size=len(example["metadata"]))
        bev_features, _ = self.extract_feat(data)
        if not return_loss and not kwargs:
            # inference on the HIP dense path: regression heads only at the cells predict() ranks
            plan = self.bbox_head.lazy_plan(self.test_cfg) if bev_features[0].dtype == torch.bfloat16 else None
            if plan is not None:
                return self.bbox_head.predict_lazy(example, bev_features, self.test_cfg, plan)
        preds = self.bbox_head(bev_features)
        if return_loss:
            return self.bbox_head.loss(example, preds, self.train_cfg)
        return self.bbox_head.predict(example, preds, self.test_cfg, **kwargs)

    @torch.no_grad()
    def forward_deferred(self, example):
        """inference with the frame's host sync postponed: queues the frame's kernels and returns a handle
        whose result() waits for this frame only and returns what forward(example, return_loss=False) does.
        A loop that queues frame i+1 before it takes frame i's result keeps the GPU busy while the host
        hands a frame over (dist_test.py's loop waits per batch)."""
        from .bbox_heads import ReadyDetections
        data = dict(points=example["points"], batch_size=len(example["metadata"]))
        bev_features, _ = self.extract_feat(data)
        plan = self.bbox_head.lazy_plan(self.test_cfg) if bev_features[0].dtype == torch.bfloat16 else None
        if plan is not None:
            out = self.bbox_head.predict_lazy(example, bev_features, self.test_cfg, plan, deferred=True)
            return out if hasattr(out, "result") else ReadyDetections(out)
        return ReadyDetections(self.bbox_head.predict(example, self.bbox_head(bev_features), self.test_cfg))

    def forward_two_stage(self, example, return_loss=True, **kwargs):
        data = dict(points=example["points"], batch_size=len(example["metadata"]))
        bev_features, backbone_features = self.extract_feat(data)
        if not return_loss and not kwargs and bev_features[0].dtype == torch.bfloat16:
            plan = self.bbox_head.lazy_plan(self.test_cfg)    # as in forward(): regression heads at the ranked cells
            if plan is not None:
                boxes = self.bbox_head.predict_lazy(example, bev_features, self.test_cfg, plan)
                return boxes, bev_features, backbone_features, None
        preds = self.bbox_head(bev_features)
        # (the reference's "manual deepcopy"; in inference the packed head output is passed through as is)
        detached = [{k: v.detach() for k, v in p.items()} for p in preds] if return_loss else preds
        boxes = self.bbox_head.predict(example, detached, self.test_cfg)
        loss = self.bbox_head.loss(example, preds, self.train_cfg) if return_loss else None
        return boxes, bev_features, backbone_features, loss
