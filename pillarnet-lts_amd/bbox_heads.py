"""HEADS registry: `CenterHead` + `SepHead` (det3d/models/bbox_heads/center_head.py:14-127).

forward(): the reference's dense 3x3 conv stacks, run by PyTorch-ROCm (MIOpen).
predict(): center_head.py:217-413 (decode, class max, score/range mask, per-task rotated NMS, task
merge) is ONE call into the HIP library (`pn_centerhead_detect`): no boolean-index syncs, no host
NMS loop; the only host sync is reading the per-sample box counts at the very end.
"""
import copy
import ctypes
import logging
from typing import List

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .nn_utils import Sequential, kaiming_init
from .registry import HEADS


class SepHead(nn.Module):
    """center_head.py:14-51: one small conv tower per regression target + the heat map."""

    def __init__(self, in_channels, heads, head_conv=64, init_bias=-2.19, **kwargs):
        super().__init__(**kwargs)
        self.heads = heads
        for head, (classes, num_conv) in heads.items():
            fc = Sequential()
            for _ in range(num_conv - 1):
                fc.add(nn.Conv2d(in_channels, head_conv, 3, stride=1, padding=1, bias=True))
                fc.add(nn.BatchNorm2d(head_conv, momentum=0.01, eps=1e-3))
                fc.add(nn.ReLU())
            fc.add(nn.Conv2d(head_conv, classes, 3, stride=1, padding=1, bias=True))
            if "hm" in head:
                fc[-1].bias.data.fill_(init_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        kaiming_init(m)
            setattr(self, head, fc)

    def forward(self, x):
        out = {}
        for head in self.heads:
            fc = getattr(self, head)
            last = fc[-1]
            y = x
            for m in list(fc._modules.values())[:-1]:
                y = m(y)
            # the output conv may be kept in fp32 while the tower runs in bf16 (to_inference):
            # bf16 heat-map logits collapse thousands of cells onto identical scores
            if y.dtype != last.weight.dtype:
                y = y.to(last.weight.dtype)
            out[head] = last(y)
        return out

    def keep_output_convs_fp32(self):
        for head in self.heads:
            getattr(self, head)[-1].float()


# channel order pn_centerhead_detect expects (include/pillarnet_hip.h)
_ORDER = ("reg", "height", "dim", "rot", "vel", "iou", "hm")


class PackedPreds(dict):
    """A task's head outputs as the usual {name: [B,c,H,W]} dict whose tensors are channel views of
    one packed fp32 map `packed` [B, H*W, ld] (channel order _ORDER) -- predict() reads that map
    directly instead of concatenating."""
    packed = None
    channels = 0


@HEADS.register_module
class CenterHead(nn.Module):
    _LEGACY_KEYS = ("dataset", "weight", "share_conv_channel", "dcn_head", "iou_reg", "order_class_names",
                    "init_bias", "num_hm_conv")

    def __init__(self, tasks: List[dict], in_channels: List, code_weights: List, common_heads=dict(),
                 logger=None, share_channel=64, reg_iou=None, pillar_size=0.1,
                 point_cloud_range=[-75.2, -75.2, -2, 75.2, 75.2, 4], **legacy):
        from_legacy = bool(legacy) or isinstance(in_channels, int) or any("stride" not in t for t in tasks)
        if from_legacy:
            # kwargs of the legacy configs (configs/nusc/**, configs/waymo/**): see legacy.legacy_head_kwargs
            unknown = set(legacy) - set(self._LEGACY_KEYS)
            if unknown:
                raise TypeError(f"CenterHead.__init__() got unexpected keyword arguments {sorted(unknown)}")
            from .legacy import legacy_head_kwargs
            kw = legacy_head_kwargs(dict(legacy, tasks=tasks, in_channels=in_channels, code_weights=code_weights,
                                         common_heads=common_heads, logger=logger, share_channel=share_channel,
                                         reg_iou=reg_iou, pillar_size=pillar_size,
                                         point_cloud_range=point_cloud_range))
            tasks, in_channels, share_channel, reg_iou = kw["tasks"], kw["in_channels"], kw["share_channel"], kw["reg_iou"]
        super().__init__()
        self.legacy_geometry = from_legacy       # stride / pillar size / range then come from test_cfg
        self.num_classes = [len(t["class_names"]) for t in tasks]
        self.class_names = [t["class_names"] for t in tasks]
        self.task_strides = [int(t["stride"]) for t in tasks]
        self.code_weights = code_weights
        self.pillar_size = pillar_size
        self.point_cloud_range = point_cloud_range
        # feature maps arrive ordered by descending stride (center_head.py:75-79)
        strides = sorted(set(self.task_strides), reverse=True)
        assert len(in_channels) == len(strides), (in_channels, strides)
        self.task_idx = [strides.index(s) for s in self.task_strides]
        self.use_iou = "iou" in common_heads
        self.use_reg_iou = reg_iou is not None
        self.reg_iou = reg_iou
        self.box_n_dim = 9 if "vel" in common_heads else 7
        self.use_direction_classifier = False
        self.logger = logger or logging.getLogger("CenterHead")
        self.logger.info(f"num_classes: {self.num_classes}")
        self.share_convs = nn.ModuleList(
            nn.Sequential(nn.Conv2d(c, share_channel, 3, padding=1, bias=True),
                          nn.BatchNorm2d(share_channel, momentum=0.01, eps=1e-3), nn.ReLU())
            for c in in_channels)
        self.task_heads = nn.ModuleList()
        for n in self.num_classes:
            heads = copy.deepcopy(common_heads)
            heads.update(dict(hm=(n, 2)))
            self.task_heads.append(SepHead(share_channel, heads))
        self.logger.info("Finish CenterHead Initialization")

    def forward(self, x):
        assert len(x) == len(self.share_convs)
        if getattr(self, "_dense_hip", False) and not self.training and x[0].dtype == torch.bfloat16:
            return self._forward_hip(x)
        shared = [conv(x[k]) for k, conv in enumerate(self.share_convs)]
        return [head(shared[self.task_idx[k]]) for k, head in enumerate(self.task_heads)]

    def _forward_hip(self, x):
        """Shared conv, ALL first-layer towers of a level as one conv, one fp32-output conv per task
        (block-diagonal weights): 2 + num_tasks launches instead of 73 convs + BN + ReLU passes."""
        from . import dense_hip
        plan = dense_hip.PLANS.get(self, dense_hip.HeadPlan)
        outs = plan.run([dense_hip.Rows.from_nchw(f) for f in x])
        preds = []
        for t, (m, nout, npad, B, H, W) in enumerate(outs):
            th = self.task_heads[t]
            d = PackedPreds()
            d.packed, d.channels = m.view(B, H * W, npad), nout
            c = 0
            img = m.view(B, H, W, npad)
            for name in _ORDER:
                if name in th.heads:
                    k = th.heads[name][0]
                    d[name] = img[..., c:c + k].permute(0, 3, 1, 2)
                    c += k
            preds.append(d)
        return preds

    def loss(self, example, preds_dicts, train_cfg, **kwargs):
        from .losses import center_head_loss
        return center_head_loss(self, example, preds_dicts, train_cfg)

    # ---------------------------------------------------------------------------------------------
    @staticmethod
    def _task_param(value, task_id, cls=None):
        """test_cfg entries are scalars, or per-task lists (after set_by_task_cfg) of per-class
        lists; returns the entry for (task, class)."""
        if isinstance(value, (list, tuple)):
            value = value[task_id]
        if cls is not None and isinstance(value, (list, tuple)):
            value = value[cls]
        return value

    @staticmethod
    def merge_double_flip(preds_dicts):
        """center_head.py:233-330, the double-flip branch: every sample arrives as four views (original, y -> -y,
        x -> -x, both); the maps are flipped back, sign-corrected and averaged -- heat map after the sigmoid, size
        after the exp, IoU after its clamp.  Returned in the *raw* parameterisation the decode kernel reads (logit
        of the mean probability, log of the mean size, 2 i - 1), so the fused decode needs no second code path."""
        merged = []
        for pd in preds_dicts:
            n = pd["hm"].shape[0]
            assert n % 4 == 0, n
            v = {}
            for k, t in pd.items():
                t = t.float().reshape(n // 4, 4, *t.shape[1:]).clone()       # [B, 4, C, H, W]
                t[:, 1] = torch.flip(t[:, 1], dims=[2])
                t[:, 2] = torch.flip(t[:, 2], dims=[3])
                t[:, 3] = torch.flip(t[:, 3], dims=[2, 3])
                v[k] = t
            out = {}
            p = torch.sigmoid(v["hm"]).mean(dim=1)
            out["hm"] = torch.log(p) - torch.log1p(-p)
            out["dim"] = torch.log(torch.exp(v["dim"].clamp(min=-1.2, max=3.2)).mean(dim=1))
            out["height"] = v["height"].mean(dim=1)
            if "iou" in v:
                out["iou"] = 2.0 * torch.clamp((v["iou"] + 1) * 0.5, min=0, max=1.0).mean(dim=1) - 1.0
            reg = v["reg"]                                   # channels (x, y)
            reg[:, 1, 1] = 1 - reg[:, 1, 1]
            reg[:, 2, 0] = 1 - reg[:, 2, 0]
            reg[:, 3] = 1 - reg[:, 3]
            out["reg"] = reg.mean(dim=1)
            rot = v["rot"]                                   # channels (sin, cos)
            rot[:, 1, 1] *= -1
            rot[:, 2, 0] *= -1
            rot[:, 3] *= -1
            out["rot"] = rot.mean(dim=1)
            if "vel" in v:
                vel = v["vel"]
                vel[:, 1, 1] *= -1
                vel[:, 2, 0] *= -1
                vel[:, 3] *= -1
                out["vel"] = vel.mean(dim=1)
            merged.append(out)
        return merged

    # ---- lazy regression heads (MI355X inference path) ---------------------------------------------------
    def lazy_plan(self, test_cfg):
        """The LazyHeadPlan of this head, or None when the dense path must run: training, fp32 / MIOpen
        backends, a rectified score (needs the IoU map everywhere), double flip (averages whole maps),
        circle NMS (ranks 4096 cells), or towers the plan does not cover."""
        import os
        if self.training or not getattr(self, "_dense_hip", False) or os.environ.get("PN_HEAD_DENSE"):
            return None
        if test_cfg.get("double_flip", False) or test_cfg.get("circular_nms", False):
            return None
        nms_cfg = test_cfg["nms"]
        multi = bool(nms_cfg.get("use_multi_class_nms", False))
        if not multi and not nms_cfg.get("use_rotate_nms", False):
            return None
        for t in range(len(self.task_heads)):
            for k in (range(self.num_classes[t]) if multi else [None]):
                if float(self._task_param(test_cfg["rectifier"], t, k)) != 0.0:
                    return None
        from . import dense_hip
        plan = dense_hip.LAZY_PLANS.get(self, dense_hip.LazyHeadPlan)
        return plan if plan.ok else None

    LAZY_MAX_CELLS = 262144     # pn_centerhead_select keeps a bit per map cell in LDS

    @torch.no_grad()
    def predict_lazy(self, example, x, test_cfg, plan, deferred=False, raw=False, **kwargs):
        """forward + predict in one: heat maps densely, cells ranked, regression heads at the ranked cells
        only.  Same outputs as `predict(example, self(x), test_cfg)`; falls back to exactly that when the
        library reports that the centre-range mask would have changed the ranking (status bit 2).
        deferred=True returns a `PendingDetections` whose result() does the frame's host sync; raw=True returns
        the fixed-capacity (boxes, scores, labels, counts + status) tensors with no host sync at all (None when
        the maps are too large for the ranked path)."""
        from . import dense_hip
        if any(f.shape[2] * f.shape[3] > self.LAZY_MAX_CELLS for f in x):
            return None if raw else self.predict(example, self(x), test_cfg, **kwargs)
        feats = [dense_hip.Rows.from_nchw(f) for f in x]
        dense = plan.run_hm(feats)
        nms_cfg = test_cfg["nms"]
        multi_class = bool(nms_cfg.get("use_multi_class_nms", False))
        dev = x[0].device
        B = x[0].shape[0]
        has_vel = int("vel" in self.task_heads[0].heads)
        has_iou = int("iou" in self.task_heads[0].heads)
        groups, owner = [], []            # owner[g] = (level, task index inside the level)
        label_offset, out_cap = 0, 0
        for t in range(len(self.task_heads)):
            lvl = self.task_idx[t]
            xs, hm = dense[lvl]
            p = plan.levels[lvl]
            gi = t - p["tasks"][0]
            for k in (range(self.num_classes[t]) if multi_class else [None]):
                g = _lib.pnDecodeGroup()
                g.maps = hm[:, gi].data_ptr()          # this task's columns of the [rows, G, npad_h] map
                g.height, g.width, g.channels, g.num_classes = xs.H, xs.W, self.num_classes[t], self.num_classes[t]
                g.row_stride = p["G"] * p["npad_h"]
                g.has_vel, g.has_iou = has_vel, has_iou
                g.class_filter = -1 if k is None else k
                g.label_offset = label_offset
                g.stride = self.task_strides[t]
                g.score_threshold = float(test_cfg["score_threshold"])
                g.rectifier, g.use_rectify = 0.0, 0
                g.nms_threshold = float(self._task_param(nms_cfg["nms_iou_threshold"], t, k))
                g.pre_max = int(self._task_param(nms_cfg["nms_pre_max_size"], t, k))
                g.post_max = int(self._task_param(nms_cfg["nms_post_max_size"], t, k))
                out_cap += g.post_max
                groups.append(g)
                owner.append((lvl, gi))
            label_offset += self.num_classes[t]
        Gd = len(groups)
        G = (_lib.pnDecodeGroup * Gd)(*groups)
        common = _lib.pnDecodeCommon()
        common.batch = B
        common.pillar_size = float(np.float32(self.pillar_size))
        common.x_min = float(np.float32(self.point_cloud_range[0]))
        common.y_min = float(np.float32(self.point_cloud_range[1]))
        rng = list(test_cfg.get("post_center_limit_range", []))
        common.has_range = int(len(rng) > 0)
        for i, v in enumerate(rng[:6]):
            common.range[i] = float(np.float32(v))
        L = _lib.lib()
        nd = self.box_n_dim
        cap = L.pn_centerhead_select_capacity(G, Gd)
        wsb = L.pn_decode_workspace_bytes(G, Gd, B, nd)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        # one block of rows per set (set = group * B + sample)
        nbr_tower = torch.empty((Gd * B, 9, 9 * cap), dtype=torch.int32, device=dev)
        nbr_out = torch.empty((Gd * B, 9, cap), dtype=torch.int32, device=dev)
        row_counts = torch.empty(Gd * B, dtype=torch.int32, device=dev)
        st = _lib.current_stream(dev)
        rc = L.pn_centerhead_select(G, Gd, ctypes.byref(common), nd, _lib.ptr(nbr_tower), _lib.ptr(nbr_out),
                                    _lib.ptr(row_counts), _lib.ptr(ws), wsb, st)
        _lib.check(rc, "pn_centerhead_select")
        # regression rows, level by level (the decode groups of a level are consecutive)
        keep_alive = []
        g0 = 0
        while g0 < Gd:
            lvl = owner[g0][0]
            g1 = g0
            while g1 < Gd and owner[g1][0] == lvl:
                g1 += 1
            set_task = [owner[g][1] for g in range(g0, g1) for _ in range(B)]
            reg = plan.run_reg(lvl, dense[lvl][0], set_task, nbr_tower[g0 * B:g1 * B], nbr_out[g0 * B:g1 * B],
                               row_counts[g0 * B:g1 * B], cap)
            keep_alive.append(reg)
            p = plan.levels[lvl]
            for g in range(g0, g1):
                G[g].maps = reg[(g - g0) * B].data_ptr()      # [B, cap, npad_r] of this group
                G[g].channels, G[g].row_stride = p["nout_r"], p["npad_r"]
            g0 = g1
        boxes = torch.empty((B, out_cap, nd), dtype=torch.float32, device=dev)
        scores = torch.empty((B, out_cap), dtype=torch.float32, device=dev)
        labels = torch.empty((B, out_cap), dtype=torch.int64, device=dev)
        num = torch.empty(B + 1, dtype=torch.int32, device=dev)
        rc = L.pn_centerhead_detect_selected(G, Gd, ctypes.byref(common), nd, out_cap, _lib.ptr(boxes), _lib.ptr(scores),
                                             _lib.ptr(labels), _lib.ptr(num), ctypes.c_void_p(num.data_ptr() + 4 * B),
                                             _lib.ptr(ws), wsb, st)
        _lib.check(rc, "pn_centerhead_detect_selected")
        if raw:
            return boxes, scores, labels, num
        pending = PendingDetections(self, example, (boxes, scores, labels, num),
                                    lambda: self.predict(example, self(x), test_cfg, **kwargs))
        return pending if deferred else pending.result()

    @torch.no_grad()
    def predict(self, example, preds_dicts, test_cfg, **kwargs):
        if test_cfg.get("double_flip", False):
            preds_dicts = self.merge_double_flip(preds_dicts)
            if isinstance(example, dict) and example.get("metadata"):
                example = dict(example, metadata=example["metadata"][::4])
        nms_cfg = test_cfg["nms"]
        circular = bool(test_cfg.get("circular_nms", False))     # center_head.py:378-386, before the other modes
        multi_class = bool(nms_cfg.get("use_multi_class_nms", False)) and not circular
        if not circular and not multi_class and not nms_cfg.get("use_rotate_nms", False):
            raise NotImplementedError
        dev = preds_dicts[0]["hm"].device
        if dev.type != "cuda":
            raise _lib.PillarnetHipError("CenterHead.predict: maps must live on the GPU (no CPU fallback)")
        B = preds_dicts[0]["hm"].shape[0]
        has_vel = int("vel" in preds_dicts[0])
        keep_alive, groups = [], []
        label_offset, out_cap = 0, 0
        for t, pd in enumerate(preds_dicts):
            has_iou = int("iou" in pd)
            H, W = pd["hm"].shape[2], pd["hm"].shape[3]
            if isinstance(pd, PackedPreds) and pd.packed is not None:
                m, CH, LD = pd.packed, pd.channels, pd.packed.shape[2]
            else:
                # one [B, H*W, C] fp32 map per task (tiny: C ~ 12 channels at head resolution)
                parts = [pd[k] for k in _ORDER if k in pd]
                m = torch.cat([p.float() for p in parts], dim=1).permute(0, 2, 3, 1).contiguous()
                CH = LD = m.shape[3]
            keep_alive.append(m)
            classes = range(self.num_classes[t]) if multi_class else [None]
            for k in classes:
                g = _lib.pnDecodeGroup()
                g.maps = m.data_ptr()
                g.height, g.width, g.channels, g.num_classes = H, W, CH, self.num_classes[t]
                g.row_stride = LD
                g.has_vel, g.has_iou = has_vel, has_iou
                g.class_filter = -1 if k is None else k
                g.label_offset = label_offset
                g.stride = self.task_strides[t]
                g.score_threshold = float(test_cfg["score_threshold"])
                g.rectifier = float(self._task_param(test_cfg["rectifier"], t, k))
                use_rect = test_cfg.get("use_rectify", False) if multi_class else False
                g.use_rectify = int(bool(self._task_param(use_rect, t, k))) if use_rect else 0
                g.nms_threshold = float(self._task_param(nms_cfg["nms_iou_threshold"], t, k))
                g.pre_max = int(self._task_param(nms_cfg["nms_pre_max_size"], t, k))
                g.post_max = int(self._task_param(nms_cfg["nms_post_max_size"], t, k))
                if circular:
                    # circle NMS on the raw heat-map score (`_circle_nms`, center_head.py:418-424): squared centre
                    # distance <= min_radius[task] suppresses; the reference ranks ALL cells above the score
                    # threshold, here the best 4096 per task
                    g.rectifier, g.use_rectify = 0.0, 0
                    g.nms_threshold = -float(test_cfg["min_radius"][t])
                    g.pre_max = 4096
                out_cap += g.post_max
                groups.append(g)
            label_offset += self.num_classes[t]
        G = (_lib.pnDecodeGroup * len(groups))(*groups)
        common = _lib.pnDecodeCommon()
        common.batch = B
        common.pillar_size = float(np.float32(self.pillar_size))
        common.x_min = float(np.float32(self.point_cloud_range[0]))
        common.y_min = float(np.float32(self.point_cloud_range[1]))
        rng = list(test_cfg.get("post_center_limit_range", []))
        common.has_range = int(len(rng) > 0)
        for i, v in enumerate(rng[:6]):
            common.range[i] = float(np.float32(v))
        L = _lib.lib()
        nd = self.box_n_dim
        wsb = L.pn_decode_workspace_bytes(G, len(groups), B, nd)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        boxes = torch.empty((B, out_cap, nd), dtype=torch.float32, device=dev)
        scores = torch.empty((B, out_cap), dtype=torch.float32, device=dev)
        labels = torch.empty((B, out_cap), dtype=torch.int64, device=dev)
        num = torch.empty(B + 1, dtype=torch.int32, device=dev)   # [B] counts + status word (both written)
        rc = L.pn_centerhead_detect(G, len(groups), ctypes.byref(common), nd, out_cap, _lib.ptr(boxes),
                                    _lib.ptr(scores), _lib.ptr(labels), _lib.ptr(num),
                                    ctypes.c_void_p(num.data_ptr() + 4 * B), _lib.ptr(ws), wsb,
                                    _lib.current_stream(dev))
        _lib.check(rc, "pn_centerhead_detect")
        if kwargs.get("raw", False):   # fixed-capacity tensors, no host sync
            return boxes, scores, labels, num
        counts = num.cpu().tolist()   # the frame's single device->host sync
        if counts[B] & 1:
            self.logger.warning("CenterHead.predict: more equal-score cells at the pre-NMS cut-off than the candidate list holds (8192); kept in ascending cell order")
        if counts[B] & 4:
            self.logger.warning("CenterHead.predict: circle NMS ranked the best 4096 cells of a task; more were above "
                                "score_threshold (the reference ranks all of them)")
        metas = example.get("metadata", None) if isinstance(example, dict) else None
        if not metas:
            metas = [None] * B
        out = []
        for b in range(B):
            n = counts[b]
            out.append({"box3d_lidar": boxes[b, :n], "scores": scores[b, :n],
                        "label_preds": labels[b, :n], "metadata": metas[b]})
        return out


class PendingDetections:
    """Detections of a frame whose kernels are queued but not waited for.

    The per-sample counts travel to pinned host memory behind the frame's last kernel and an event marks
    that copy; result() waits for the event alone, so a caller may queue the next frame before it asks for
    this one and the GPU never idles through the host's hand-over between frames."""

    def __init__(self, head, example, tensors, redo):
        self.head, self.example, self.redo = head, example, redo     # redo(): the frame again on the dense path
        self.boxes, self.scores, self.labels, num = tensors
        self.host = torch.empty(num.shape, dtype=num.dtype, pin_memory=True)
        self.host.copy_(num, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self.out = None

    def result(self):
        if self.out is not None:
            return self.out
        self.event.synchronize()          # the frame's single device->host sync
        head, example = self.head, self.example
        B = self.boxes.shape[0]
        counts = self.host.tolist()
        if counts[B] & 2:                 # the range mask removed ranked cells of an over-full set: dense redo
            self.out, self.redo = self.redo(), None
            return self.out
        if counts[B] & 1:
            head.logger.warning("CenterHead.predict: more equal-score cells at the pre-NMS cut-off than the candidate list holds (8192); kept in ascending cell order")
        if counts[B] & 4:
            head.logger.warning("CenterHead.predict: circle NMS ranked the best 4096 cells of a task; more were above "
                                "score_threshold (the reference ranks all of them)")
        metas = example.get("metadata", None) if isinstance(example, dict) else None
        if not metas:
            metas = [None] * B
        self.out = [{"box3d_lidar": self.boxes[b, :counts[b]], "scores": self.scores[b, :counts[b]],
                     "label_preds": self.labels[b, :counts[b]], "metadata": metas[b]} for b in range(B)]
        self.redo = None
        return self.out


class ReadyDetections:
    """a finished result behind the PendingDetections interface (paths that sync inside the call)."""

    def __init__(self, out):
        self.out = out

    def result(self):
        return self.out


def resolve_legacy_geometry(head, test_cfg):
    """fill a legacy head's stride / pillar size / range from test_cfg (out_size_factor, voxel_size, pc_range)."""
    if test_cfg is None or not getattr(head, "legacy_geometry", False):
        return
    osf = int(test_cfg.get("out_size_factor", 8))
    head.task_strides = [osf if s == 0 else s for s in head.task_strides]
    vs = test_cfg.get("voxel_size", None)
    if vs is not None:
        head.pillar_size = float(vs[0])
    rng = test_cfg.get("pc_range", None)
    if rng is not None and len(rng) >= 2:
        head.point_cloud_range = list(rng[:2]) + list(head.point_cloud_range[2:])
