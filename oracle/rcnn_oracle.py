"""CPU restatement of the Pillar R-CNN second stage.  TEST INFRASTRUCTURE ONLY (checker for tests/).

Follows (reference paths):
  det3d/core/utils/center_utils.py  center_to_grid_box2d / bilinear_interpolate_torch
  det3d/core/bbox/box_torch_ops.py  rotation_2d, rotate_points_along_z (:120-138)
  det3d/models/second_stage/bev_interpolation.py:162-308   BEVStrideFeature
  det3d/models/roi_heads/roi_mix_head.py:17-122, roi_head_template.py:189-219
  det3d/models/detectors/pillar_rcnn.py:133-166            post_process
Pinned by tests/golden/rcnn.npz, produced by the reference's own modules (make_golden.golden_rcnn).
"""
import numpy as np
import torch
import torch.nn.functional as Fn

F32 = np.float32
EPS = 1e-3


def _t(sd, k):
    return torch.from_numpy(np.asarray(sd[k]))


def _bn(sd, p, x):
    return Fn.batch_norm(x, _t(sd, p + ".running_mean"), _t(sd, p + ".running_var"), _t(sd, p + ".weight"),
                         _t(sd, p + ".bias"), False, 0.0, EPS)


def grid_points(rois, grid):
    """[N,7] -> [N, G*G, 2] world xy of the rotated G x G lattice (cell centres of the RoI)."""
    rois = np.asarray(rois, F32)
    ii, jj = np.meshgrid(np.arange(grid), np.arange(grid), indexing="ij")
    idx = np.stack([ii.ravel(), jj.ravel()], -1).astype(F32)                      # (i, j)
    dims = rois[:, None, 3:5]
    local = (idx[None] + F32(0.5)) / F32(grid) * dims - dims / F32(2)
    c, s = np.cos(rois[:, 6]).astype(F32), np.sin(rois[:, 6]).astype(F32)
    x = local[..., 0] * c[:, None] + local[..., 1] * s[:, None]
    y = -local[..., 0] * s[:, None] + local[..., 1] * c[:, None]
    return np.stack([x + rois[:, None, 0], y + rois[:, None, 1]], -1).astype(F32)


def bilinear(fmap_hwc, x, y):
    H, W = fmap_hwc.shape[:2]
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    x0, x1 = np.clip(x0, 0, W - 1), np.clip(x1, 0, W - 1)
    y0, y1 = np.clip(y0, 0, H - 1), np.clip(y1, 0, H - 1)
    wa = (x1.astype(F32) - x) * (y1.astype(F32) - y)
    wb = (x1.astype(F32) - x) * (y - y0.astype(F32))
    wc = (x - x0.astype(F32)) * (y1.astype(F32) - y)
    wd = (x - x0.astype(F32)) * (y - y0.astype(F32))
    return (fmap_hwc[y0, x0] * wa[:, None] + fmap_hwc[y1, x0] * wb[:, None] + fmap_hwc[y0, x1] * wc[:, None]
            + fmap_hwc[y1, x1] * wd[:, None]).astype(F32)


def roi_grid_pool(rois, fmap, grid, cell, x_min, y_min):
    """rois [B,R,7], fmap [B,C,H,W] -> ([B,R,G*G,C], [B,R,G*G,2])."""
    B, R = rois.shape[:2]
    pts = grid_points(rois.reshape(B * R, -1), grid).reshape(B, R, grid * grid, 2)
    out = np.zeros((B, R, grid * grid, fmap.shape[1]), F32)
    for b in range(B):
        hwc = np.ascontiguousarray(np.transpose(fmap[b], (1, 2, 0)))
        p = pts[b].reshape(-1, 2)
        xi = (p[:, 0] - F32(x_min)) / F32(cell)
        yi = (p[:, 1] - F32(y_min)) / F32(cell)
        out[b] = bilinear(hwc, xi.astype(F32), yi.astype(F32)).reshape(R, grid * grid, -1)
    return out, pts


def bev_stride_feature(sd, bev_feature, laterals, p="second_stage.0"):
    def up(name, x):
        w = _t(sd, f"{p}.{name}.0.weight")
        return torch.relu(_bn(sd, f"{p}.{name}.1", Fn.conv_transpose2d(x, w, stride=w.shape[-1])))
    feats = [up("top_down_conv", torch.from_numpy(bev_feature))]
    feats += [up(f"lat_conv.{k}", torch.from_numpy(x)) for k, x in enumerate(laterals)]
    y = Fn.conv2d(torch.cat(feats, 1), _t(sd, f"{p}.fusion_conv.0.weight"), _t(sd, f"{p}.fusion_conv.0.bias"), padding=1)
    return torch.relu(_bn(sd, f"{p}.fusion_conv.1", y)).numpy()


def _fc(sd, p, x, n_hidden, dropout_slots):
    """Conv1d(k=1)+BN+ReLU stacks; `dropout_slots` = Sequential indices the (eval-inert) Dropouts occupy."""
    i = 0
    for _ in range(n_hidden):
        x = torch.relu(_bn(sd, f"{p}.{i + 1}", Fn.conv1d(x, _t(sd, f"{p}.{i}.weight"))))
        i += 3
        if i in dropout_slots:
            i += 1
    return x, i


def roi_head(sd, roi_features, rois, shared=2, cls=2, reg=2, dp=True, p="roi_head"):
    B, R = rois.shape[:2]
    x = torch.from_numpy(roi_features).reshape(B * R, -1, 1)
    x, _ = _fc(sd, p + ".shared_fc_layer", x, shared, {3} if dp and shared > 1 else set())
    hc, i = _fc(sd, p + ".cls_layers", x, cls, {3} if dp else set())
    c = Fn.conv1d(hc, _t(sd, f"{p}.cls_layers.{i}.weight"), _t(sd, f"{p}.cls_layers.{i}.bias"))
    hr, i = _fc(sd, p + ".reg_layers", x, reg, {3} if dp else set())
    r = Fn.conv1d(hr, _t(sd, f"{p}.reg_layers.{i}.weight"), _t(sd, f"{p}.reg_layers.{i}.bias"))
    c, r = c.squeeze(-1).numpy(), r.squeeze(-1).numpy()
    flat = rois.reshape(B * R, -1).astype(F32)
    boxes = r + np.concatenate([np.zeros((B * R, 3), F32), flat[:, 3:]], 1)
    cs, sn = np.cos(flat[:, 6]), np.sin(flat[:, 6])
    x_, y_ = boxes[:, 0] * cs + boxes[:, 1] * sn, -boxes[:, 0] * sn + boxes[:, 1] * cs   # row vector @ [[c,-s],[s,c]]
    boxes = boxes.copy()
    boxes[:, 0], boxes[:, 1] = x_ + flat[:, 0], y_ + flat[:, 1]
    boxes[:, 2] += flat[:, 2]
    return c.reshape(B, R, -1), boxes.reshape(B, R, -1).astype(F32)


def post_process(cls, boxes, roi_scores, roi_labels):
    out = []
    for i in range(len(boxes)):
        score = np.sqrt((1.0 / (1.0 + np.exp(-cls[i].astype(F32)))).reshape(-1) * roi_scores[i].reshape(-1)).astype(F32)
        m = (roi_labels[i] != 0) & (boxes[i][:, 3:6] > 0).all(1)
        out.append(dict(box3d_lidar=boxes[i][m], scores=score[m], label_preds=roi_labels[i][m] - 1))
    return out
