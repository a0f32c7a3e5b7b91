"""CenterHead training losses (det3d/models/losses/centernet_loss.py:9-127, det3d/core/utils/
center_utils.py:122-226) and `CenterHead.loss` (det3d/models/bbox_heads/center_head.py:133-214).
Plain torch: these are tiny gathers over <= max_objs targets, not a hot path.  Nothing here reads a value back
to the host (the reference branches on `mask.sum() == 0`, indexes with boolean masks and moves its log values to
the CPU per task, ~70 host waits per step): empty tasks fall out of the arithmetic, masked rows are computed on
stand-in boxes and dropped with `where`, and the log values stay on the device until somebody asks for them."""
from collections import defaultdict

import torch
import torch.nn.functional as F


def gather_at(feat_nhwc, ind):
    """[B,H,W,C] map sampled at flat cell ids ind [B,K] -> [B,K,C]  (center_utils.py:75-78)."""
    B, C = feat_nhwc.shape[0], feat_nhwc.shape[-1]
    flat = feat_nhwc.reshape(B, -1, C)
    return flat.gather(1, ind.unsqueeze(2).expand(B, ind.shape[1], C))


def fast_focal_loss(out, target, ind, mask, cat):
    """centernet_loss.py:34-63.  out/target [B,H,W,C] (out already sigmoid + clamp)."""
    mask = mask.float()
    neg = (torch.log(1 - out) * out.pow(2) * (1 - target).pow(4)).sum()
    pos_pred = gather_at(out, ind).gather(2, cat.unsqueeze(2))            # [B,K,1]
    pos = (torch.log(pos_pred) * (1 - pos_pred).pow(2) * mask.unsqueeze(2)).sum()
    num_pos = mask.sum()
    return -(pos + neg) / num_pos.clamp(min=1.0)      # no positives: pos == 0 and the reference returns -neg


def reg_loss(output, mask, ind, target):
    """centernet_loss.py:9-31: masked L1 per box-code element, normalised by the object count."""
    pred = gather_at(output, ind)        # (no objects: every term is 0, as the reference's early return)
    m = mask.float().unsqueeze(2)
    loss = F.l1_loss(pred * m, target * m, reduction="none") / (m.sum() + 1e-4)
    return loss.transpose(2, 0).sum(dim=2).sum(dim=1)


def _corners2d(center, dim):
    norm = torch.tensor([[-0.5, -0.5], [-0.5, 0.5], [0.5, 0.5], [0.5, -0.5]], dtype=torch.float32, device=dim.device)
    return dim.view(-1, 1, 2) * norm.view(1, 4, 2) + center.view(-1, 1, 2)


def axis_aligned_iou3d(pred, gt, kind="IoU"):
    """bbox3d_overlaps_{iou,giou,diou} (center_utils.py:131-226): heading is ignored."""
    assert pred.shape[0] == gt.shape[0]
    q, g = _corners2d(pred[:, :2], pred[:, 3:5]), _corners2d(gt[:, :2], gt[:, 3:5])
    inter_xy = torch.clamp(torch.minimum(q[:, 2], g[:, 2]) - torch.maximum(q[:, 0], g[:, 0]), min=0)
    top = lambda b: b[:, 2] + 0.5 * b[:, 5]
    bot = lambda b: b[:, 2] - 0.5 * b[:, 5]
    inter_h = torch.clamp(torch.minimum(top(gt), top(pred)) - torch.maximum(bot(gt), bot(pred)), min=0)
    v_inter = inter_xy[:, 0] * inter_xy[:, 1] * inter_h
    v_union = gt[:, 3] * gt[:, 4] * gt[:, 5] + pred[:, 3] * pred[:, 4] * pred[:, 5] - v_inter
    if kind == "IoU":
        return torch.clamp(v_inter / v_union, min=0, max=1.0)
    outer_xy = torch.clamp(torch.maximum(q[:, 2], g[:, 2]) - torch.minimum(q[:, 0], g[:, 0]), min=0)
    outer_h = torch.clamp(torch.maximum(top(gt), top(pred)) - torch.minimum(bot(gt), bot(pred)), min=0)
    if kind == "GIoU":
        closure = outer_xy[:, 0] * outer_xy[:, 1] * outer_h
        return torch.clamp(v_inter / v_union - (closure - v_union) / closure, min=-1.0, max=1.0)
    if kind == "DIoU":
        inter_diag = (gt[:, 0:3] - pred[:, 0:3]).pow(2).sum(-1)
        outer_diag = outer_xy[:, 0] ** 2 + outer_xy[:, 1] ** 2 + outer_h ** 2
        return torch.clamp(v_inter / v_union - inter_diag / outer_diag, min=-1.0, max=1.0)
    raise NotImplementedError(kind)


def _masked_boxes(boxes, m):
    """rows outside the mask replaced by a unit box at the origin: their IoU is finite, so neither the value nor
    the gradient of a `where` over the result can pick up a NaN from an all-zero padding row."""
    unit = boxes.new_zeros(boxes.shape[-1])
    unit[3:6] = 1.0
    return torch.where(m.unsqueeze(-1), boxes, unit)


def iou_reg_loss(box_pred, mask, ind, box_gt, kind):
    """centernet_loss.py:94-127."""
    m = mask.bool()
    B, K = m.shape
    pred, gt = _masked_boxes(gather_at(box_pred, ind), m), _masked_boxes(box_gt, m)
    iou = axis_aligned_iou3d(pred.reshape(B * K, -1), gt.reshape(B * K, -1), kind)
    return torch.where(m.reshape(-1), 1.0 - iou, iou.new_zeros(())).sum() / (m.sum() + 1e-4)


def iou_loss(iou_pred, mask, ind, box_pred, box_gt):
    """centernet_loss.py:66-91: L1 between the predicted IoU and 2*IoU3D(rotated)-1."""
    from .iou3d_nms import boxes_aligned_iou3d_gpu
    m = mask.bool()
    B, K = m.shape
    pred = gather_at(iou_pred, ind).reshape(B * K, -1)
    boxes, gt = _masked_boxes(gather_at(box_pred, ind), m), _masked_boxes(box_gt, m)
    target = 2 * boxes_aligned_iou3d_gpu(boxes.reshape(B * K, -1).contiguous(), gt.reshape(B * K, -1).contiguous()) - 1
    err = (pred - target.reshape(pred.shape)).abs()
    return torch.where(m.reshape(-1, 1), err, err.new_zeros(())).sum() / (m.sum() + 1e-4)


def center_head_loss(head, example, preds_dicts, train_cfg):
    """CenterHead.loss (center_head.py:133-214): dict of per-task lists."""
    merged = defaultdict(list)
    for t, pd in enumerate(preds_dicts):
        pd = {k: v.permute(0, 2, 3, 1).contiguous().float() for k, v in pd.items()}
        hm = torch.clamp(torch.sigmoid(pd["hm"]), min=1e-4, max=1 - 1e-4)
        mask, ind = example["mask"][t], example["ind"][t]
        hm_loss = fast_focal_loss(hm, example["hm"][t], ind, mask, example["cat"][t])
        target_box = example["anno_box"][t]
        if "vel" in pd:
            anno = torch.cat((pd["reg"], pd["height"], pd["dim"], pd["vel"], pd["rot"]), dim=-1)
        else:
            anno = torch.cat((pd["reg"], pd["height"], pd["dim"], pd["rot"]), dim=-1)
            target_box = target_box[..., [0, 1, 2, 3, 4, 5, -2, -1]]
        box_loss = reg_loss(anno, mask, ind, target_box)
        loc_loss = (box_loss * box_loss.new_tensor(head.code_weights)).sum()
        loss = hm_loss * train_cfg["hm_weight"] + loc_loss * train_cfg["bbox_weight"]
        ret = {"hm_loss": hm_loss.detach(), "loc_loss": loc_loss, "loc_loss_elem": box_loss.detach(),
               "num_positive": mask.float().sum()}
        if head.use_iou or head.use_reg_iou:
            B, H, W, _ = pd["dim"].shape
            ys, xs = torch.meshgrid(torch.arange(H, device=hm.device), torch.arange(W, device=hm.device), indexing="ij")
            xs = xs.view(1, H, W, 1).float() + pd["reg"][..., 0:1]
            ys = ys.view(1, H, W, 1).float() + pd["reg"][..., 1:2]
            xs = xs * head.task_strides[t] * head.pillar_size + head.point_cloud_range[0]
            ys = ys * head.task_strides[t] * head.pillar_size + head.point_cloud_range[1]
            rot = torch.atan2(pd["rot"][..., 0:1], pd["rot"][..., 1:2])
            boxes = torch.cat([xs, ys, pd["height"], torch.exp(pd["dim"].clamp(min=-1.2, max=3.2)), rot], dim=-1)
            if head.use_iou:
                il = iou_loss(pd["iou"], mask, ind, boxes.detach(), example["gt_box"][t])
                loss = loss + il * train_cfg["iou_weight"]
                ret["iou_loss"] = il.detach()
            if head.use_reg_iou:
                rl = iou_reg_loss(boxes, mask, ind, example["gt_box"][t], head.reg_iou)
                loss = loss + rl * train_cfg["reg_iou_weight"]
                ret["reg_iou_loss"] = rl.detach()
        ret["loss"] = loss
        for k, v in ret.items():
            merged[k].append(v)
    return merged
