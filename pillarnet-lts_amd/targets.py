"""CenterNet-style training targets: `AssignLabel` (det3d/datasets/pipelines/preprocess.py:178-350)
and its gaussian helpers (det3d/core/utils/center_utils.py:16-63), restated in numpy.  Used to make
synthetic training examples of the real shapes (no dataset is reachable offline)."""
import numpy as np


def gaussian_radius(det_size, min_overlap=0.5):
    """center_utils.py:16-38: the smallest of the three corner-shift radii that keep IoU >= min_overlap."""
    h, w = det_size
    r = []
    for a, b, c in ((1.0, h + w, w * h * (1 - min_overlap) / (1 + min_overlap)),
                    (4.0, 2 * (h + w), (1 - min_overlap) * w * h),
                    (4 * min_overlap, -2 * min_overlap * (h + w), (min_overlap - 1) * w * h)):
        r.append((b + np.sqrt(b * b - 4 * a * c)) / 2)
    return min(r)


def gaussian2d(shape, sigma=1.0):
    m, n = [(s - 1.0) / 2.0 for s in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    g = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    g[g < np.finfo(g.dtype).eps * g.max()] = 0
    return g


def draw_gaussian(heatmap, center, radius, k=1):
    """center_utils.py:49-63 (draw_umich_gaussian): element-wise max with a clipped gaussian patch."""
    d = 2 * radius + 1
    g = gaussian2d((d, d), sigma=d / 6)
    x, y = int(center[0]), int(center[1])
    H, W = heatmap.shape[:2]
    left, right = min(x, radius), min(W - x, radius + 1)
    top, bottom = min(y, radius), min(H - y, radius + 1)
    hm = heatmap[y - top:y + bottom, x - left:x + right]
    gg = g[radius - top:radius + bottom, radius - left:radius + right]
    if min(gg.shape) > 0 and min(hm.shape) > 0:
        np.maximum(hm, gg * k, out=hm)
    return heatmap


def limit_period(val, offset=0.5, period=np.pi):
    return val - np.floor(val / period + offset) * period


def assign_labels(gt_boxes, gt_classes, tasks, pc_range, pillar_size, gaussian_overlap=0.1, max_objs=500,
                  min_radius=2, dataset="NuScenesDataset"):
    """gt_boxes [G, 9] (x,y,z,w,l,h,vx,vy,rot), gt_classes [G] 1-based over the concatenated class list.
    Returns dict(hm, anno_box, ind, mask, cat, gt_box): per-task lists, as `AssignLabel` emits."""
    pc_range = np.asarray(pc_range, np.float32)
    ps = np.float32(pillar_size)
    grid = np.round((pc_range[3:5] - pc_range[:2]) / ps).astype(np.int64)
    out = dict(hm=[], anno_box=[], ind=[], mask=[], cat=[], gt_box=[])
    flag = 0
    flat_boxes, flat_cls = [], []      # "used for two stage code", preprocess.py:324-343
    for task in tasks:
        names, stride = task["class_names"], int(task["stride"])
        sel = [np.where(gt_classes == i + 1 + flag)[0] for i in range(len(names))]
        boxes = np.concatenate([gt_boxes[s] for s in sel], axis=0).astype(np.float32).copy()
        cls = np.concatenate([gt_classes[s] - flag for s in sel])
        boxes[:, -1] = limit_period(boxes[:, -1], offset=0.5, period=np.pi * 2)
        flat_boxes.append(boxes)
        flat_cls.append(cls + flag)
        g = grid // stride
        hm = np.zeros((len(names), g[1], g[0]), np.float32)
        anno = np.zeros((max_objs, 10), np.float32)
        gtb = np.zeros((max_objs, 7), np.float32)
        ind = np.zeros(max_objs, np.int64)
        mask = np.zeros(max_objs, np.uint8)
        cat = np.zeros(max_objs, np.int64)
        for k in range(min(boxes.shape[0], max_objs)):
            cid = int(cls[k]) - 1
            w, l = boxes[k, 3] / (ps * stride), boxes[k, 4] / (ps * stride)
            if not (w > 0 and l > 0):
                continue
            mr = min_radius[cid] if isinstance(min_radius, (list, tuple)) else min_radius
            radius = max(mr, int(gaussian_radius((l, w), min_overlap=gaussian_overlap)))
            ct = np.array([(boxes[k, 0] - pc_range[0]) / (ps * stride), (boxes[k, 1] - pc_range[1]) / (ps * stride)],
                          dtype=np.float32)
            ci = ct.astype(np.int32)
            if not (0 <= ci[0] < g[0] and 0 <= ci[1] < g[1]):
                continue
            draw_gaussian(hm[cid], ct, radius)
            cat[k], ind[k], mask[k] = cid, ci[1] * g[0] + ci[0], 1
            gtb[k] = boxes[k, [0, 1, 2, 3, 4, 5, 8]]
            rot = boxes[k, 8]
            anno[k] = np.concatenate((ct - ci, boxes[k, 2], np.log(boxes[k, 3:6]), boxes[k, 6], boxes[k, 7],
                                      np.sin(rot), np.cos(rot)), axis=None)
        out["hm"].append(hm.transpose(1, 2, 0))
        out["anno_box"].append(anno)
        out["ind"].append(ind)
        out["mask"].append(mask)
        out["cat"].append(cat)
        out["gt_box"].append(gtb)
        flag += len(names)
    # x, y, z, w, l, h, rotation_y, velocity_x, velocity_y, class (1-based over all tasks); zero rows pad
    both = np.concatenate((np.concatenate(flat_boxes, 0), np.concatenate(flat_cls).reshape(-1, 1).astype(np.float32)), 1)
    assert len(both) <= max_objs
    gbc = np.zeros((max_objs, 10), np.float32)
    gbc[: len(both)] = both[:, [0, 1, 2, 3, 4, 5, 8, 6, 7, 9]]
    out["gt_boxes_and_cls"] = gbc
    return out


def synthetic_example(points, tasks, pc_range, pillar_size, n_boxes=30, seed=0, train_cfg=None):
    """One synthetic training sample: random boxes inside the range, classes uniform over all tasks."""
    rng = np.random.default_rng(seed)
    ncls = sum(len(t["class_names"]) for t in tasks)
    ext = (pc_range[3] - pc_range[0]) / 2 * 0.9
    cx = (pc_range[0] + pc_range[3]) / 2
    cy = (pc_range[1] + pc_range[4]) / 2
    b = np.zeros((n_boxes, 9), np.float32)
    b[:, 0] = cx + rng.uniform(-ext, ext, n_boxes)
    b[:, 1] = cy + rng.uniform(-ext, ext, n_boxes)
    b[:, 2] = rng.normal(-1.0, 0.5, n_boxes)
    b[:, 3:6] = np.exp(rng.normal(np.log([1.9, 4.5, 1.6]), 0.25, (n_boxes, 3)))
    b[:, 6:8] = rng.normal(0, 1.0, (n_boxes, 2))
    b[:, 8] = rng.uniform(-np.pi, np.pi, n_boxes)
    c = rng.integers(1, ncls + 1, n_boxes)
    a = train_cfg["assigner"] if train_cfg else {}
    t = assign_labels(b, c, tasks, pc_range, pillar_size, a.get("gaussian_overlap", 0.1), a.get("max_objs", 500),
                      a.get("min_radius", 2))
    return dict(points=points, gt_boxes=b, gt_classes=c, **t)


def collate(samples, device):
    """Batch of samples -> the example dict the detector's train forward reads (the relevant part of
    collate_kitti, det3d/torchie/parallel/collate.py:91-171): per-task tensors stacked over samples."""
    import torch
    ex = {"points": [torch.as_tensor(s["points"]).to(device) for s in samples], "metadata": [{} for _ in samples]}
    for key in ("hm", "anno_box", "ind", "mask", "cat", "gt_box"):
        ntask = len(samples[0][key])
        ex[key] = [torch.as_tensor(np.stack([s[key][t] for s in samples])).to(device) for t in range(ntask)]
    if "gt_boxes_and_cls" in samples[0]:
        ex["gt_boxes_and_cls"] = torch.as_tensor(np.stack([s["gt_boxes_and_cls"] for s in samples])).to(device)
    return ex
