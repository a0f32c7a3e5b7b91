#!/usr/bin/env python3
"""Full-size golden vectors: the CPU oracle run ONCE, in the build container, on the synthetic
frames of BASELINE configs[1] (nuScenes PillarNet-18, 1440^2, 10 x 30 000 points), configs[2]
(Waymo PillarNet-34 + RPNG, 1504^2, 180 000 points) and the first stage of configs[4] (Waymo
PillarNet-18 + RPNG under Pillar R-CNN), compacted to a few hundred KB each:

  * sha256 of the seeded weights (the GPU test rebuilds them from the same seeds and checks),
  * per sparse level: row count, sha256 of the int32 (b, y, x) rows, the occupancy bitmap of the
    coarser levels, feature rows at sampled positions,
  * conv4 / conv5 / BEV maps and every head map at sampled cells (all channels),
  * the oracle's detections (fp32 boxes, scores, labels, flat cells, min |IoU - thr| consulted).

Nothing here needs /root/reference: the oracle is this repository's own CPU restatement (whose
pinning to the reference is described in oracle/*.py).  Usage:

    python tests/golden/make_fullsize.py [nusc] [nusc1] [waymo] [rcnn]           # -> tests/golden/fullsize_*.npz
    python tests/golden/make_fullsize.py --calibrated [nusc] ...                 # -> fullsize_*_calibrated.npz
"""
import copy
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import detector_oracle as do  # noqa: E402
sys.path.insert(0, os.path.dirname(HERE))
from fullsize_cases import build_model, sha, state_sha  # noqa: E402

N_ROWS = 96      # sampled rows per sparse level
N_CELLS = 96     # sampled cells per dense map
BETA = 1.5       # calibrated model: BatchNorm bias / weight (see fullsize_cases.apply_calibration)


def compact(mid, dets, seed):
    rng = np.random.default_rng(seed)
    out = {}
    r = mid["reader"]
    M = r["indices"].shape[0]
    rows = np.sort(rng.choice(M, N_ROWS, replace=False))
    out.update(reader_count=np.int64(M), reader_idx_sha=sha(r["indices"].astype(np.int32)),
               reader_rows=rows, reader_feat=r["features"][rows].astype(np.float32),
               reader_idx_rows=r["indices"][rows].astype(np.int32))
    for name, (feats, idx, shape) in mid["levels"].items():
        n = idx.shape[0]
        rows = np.sort(rng.choice(n, N_ROWS, replace=False))
        out[f"{name}_count"] = np.int64(n)
        out[f"{name}_idx_sha"] = sha(idx.astype(np.int32))
        out[f"{name}_rows"] = rows
        out[f"{name}_idx_rows"] = idx[rows].astype(np.int32)
        out[f"{name}_feat"] = feats[rows].astype(np.float32)
        if name != "conv1":
            occ = np.zeros(shape, bool)
            occ[idx[:, 1], idx[:, 2]] = True            # batch 1
            out[f"{name}_occ"] = np.packbits(occ)
            out[f"{name}_shape"] = np.asarray(shape, np.int64)

    def dense_sample(key, arr, prefer=None):
        _, C, H, W = arr.shape
        if prefer is not None:                            # half the cells on active sites
            act = np.flatnonzero(prefer.reshape(-1))
            a = rng.choice(act, min(N_CELLS // 2, act.size), replace=False)
            cells = np.unique(np.concatenate([a, rng.choice(H * W, N_CELLS - a.size, replace=False)]))
        else:
            cells = np.sort(rng.choice(H * W, N_CELLS, replace=False))
        out[f"{key}_cells"] = cells.astype(np.int64)
        out[f"{key}_shape"] = np.asarray(arr.shape, np.int64)
        out[f"{key}_vals"] = arr[0].reshape(C, H * W)[:, cells].T.astype(np.float32).copy()

    dense_sample("dense4", mid["conv4"], prefer=np.abs(mid["conv4"][0]).sum(0) > 0)
    dense_sample("dense5", mid["conv5"])
    for k, b in enumerate(mid["bev"]):
        dense_sample(f"bev{k}", b)
    for t, pd in enumerate(mid["preds"]):
        for name, v in pd.items():
            dense_sample(f"pred{t}_{name}", v)
    d = dets[0]
    out.update(det_boxes=d["box3d_lidar"].astype(np.float32), det_scores=d["scores"].astype(np.float32),
               det_labels=d["label_preds"].astype(np.int64), det_min_gap=np.float64(d["min_gap"]),
               det_cells=np.concatenate(d["cells"]).astype(np.int64),
               det_group_sizes=np.asarray([len(c) for c in d["cells"]], np.int64))
    return out


# output statistics the calibrated model's regression heads are given (mean, std over the map): offsets inside a cell,
# height [m], log sizes, sin / cos, velocity [m/s], IoU logit
HEAD_TARGETS = {"reg": (0.5, 0.25), "height": (-1.0, 0.5), "dim": (0.5, 0.3), "rot": (0.0, 0.5), "vel": (0.0, 1.0),
                "iou": (0.0, 0.5)}


def make_calibrated(kind):
    """The same frame and seeded weights as a trained-like model (fullsize_cases.apply_calibration): BatchNorm
    statistics measured by the oracle on this frame with the units in the ReLU's linear region, output convs of the head
    mean-centred and scaled to O(1) outputs, heat maps peaked; the oracle's detections on it."""
    cfg, model, first_cfg, first, pts = build_model(kind)
    sd = {k: v.detach().float().cpu() for k, v in first.state_dict().items()}
    t0 = time.time()
    fwd = lambda: do.pillarnet_forward(sd, [pts], first_cfg, copy.deepcopy(first.test_cfg), return_intermediates=True)
    stats = do.calibrate_batchnorm(sd, [pts], first_cfg, copy.deepcopy(first.test_cfg), beta_over_gamma=BETA)
    # output convs: every output channel's weights sum to zero, so that it answers to the map's variation, not its level
    finals = {}
    for key in list(sd):
        parts = key.split(".")
        if key.startswith("bbox_head.task_heads.") and key.endswith(".weight") and key[:-7] + ".bias" in sd and \
                key[:-7] + ".running_mean" not in sd and parts[3] in list(HEAD_TARGETS) + ["hm"] and \
                not any(k.startswith(".".join(parts[:4]) + "." + str(int(parts[4]) + 1) + ".") for k in sd):
            w = sd[key].double()
            sd[key] = (w - w.mean((1, 2, 3), keepdim=True)).float()
            sd[key[:-7] + ".bias"] = torch.zeros_like(sd[key[:-7] + ".bias"])
            finals[(int(parts[2]), parts[3])] = key[:-7]
    _, mid = fwd()
    peak = []
    for t, pd in enumerate(mid["preds"]):
        for name, v in pd.items():
            p = finals[(t, name)]
            if name == "hm":
                u = v.max(1).reshape(-1).astype(np.float64)
                q0, q1 = np.quantile(u, [0.99, 0.999])
                gain = 4.0 / (q1 - q0)
                peak.append((gain, q0))
                sd[p + ".weight"] = (sd[p + ".weight"].double() * gain).float()
                sd[p + ".bias"] = torch.full_like(sd[p + ".bias"], float(-2.19 - gain * q0))
                continue
            mean, std = HEAD_TARGETS[name]
            flat = v[0].reshape(v.shape[1], -1).astype(np.float64)
            gain = std / np.maximum(flat.std(1), 1e-9)
            sd[p + ".weight"] = (sd[p + ".weight"].double() * torch.from_numpy(gain).view(-1, 1, 1, 1)).float()
            sd[p + ".bias"] = torch.from_numpy(mean - gain * flat.mean(1)).float()
    dets, mid = fwd()
    d = dets[0]
    above = [int((1 / (1 + np.exp(-pd["hm"].max(1).astype(np.float64))) > first.test_cfg["score_threshold"]).sum())
             for pd in mid["preds"]]
    out = {"set." + k: v for k, v in stats.items()}
    for p in finals.values():
        out["set." + p + ".weight"], out["set." + p + ".bias"] = sd[p + ".weight"].numpy(), sd[p + ".bias"].numpy()
    out.update(det_boxes=d["box3d_lidar"].astype(np.float32),
               det_scores=d["scores"].astype(np.float32), det_labels=d["label_preds"].astype(np.int64),
               det_min_gap=np.float64(d["min_gap"]), cells_above_threshold=np.asarray(above, np.int64),
               weights_sha=state_sha(model.state_dict()), points_sha=sha(pts))
    rng = np.random.default_rng(11)
    for k, b in enumerate(mid["bev"]):                     # a few BEV cells, to bound the bf16 map error on this model too
        _, C, H, W = b.shape
        cells = np.sort(rng.choice(H * W, N_CELLS, replace=False))
        out[f"bev{k}_cells"], out[f"bev{k}_vals"] = cells.astype(np.int64), b[0].reshape(C, H * W)[:, cells].T.astype(np.float32).copy()
    path = os.path.join(HERE, f"fullsize_{kind}_calibrated.npz")
    np.savez_compressed(path, **out)
    print(f"{kind} calibrated: oracle {time.time() - t0:.1f} s, {len(d['scores'])} detections, cells above the threshold per "
          f"task {above}, heat-map gains {[round(g, 3) for g, _ in peak]}, score quantiles "
          f"{np.quantile(d['scores'], [0, .1, .5, .9, 1]).round(3)}, min gap {d['min_gap']:.3g} -> {path} "
          f"({os.path.getsize(path) / 1024:.0f} KiB)", flush=True)


def make(kind):
    cfg, model, first_cfg, first, pts = build_model(kind)
    sd = {k: v.detach().float().cpu() for k, v in first.state_dict().items()}
    test_cfg = first.test_cfg          # regrouped per task by the detector's constructor
    t0 = time.time()
    dets, mid = do.pillarnet_forward(sd, [pts], first_cfg, copy.deepcopy(test_cfg), return_intermediates=True)
    out = compact(mid, dets, seed=7)
    out["weights_sha"] = state_sha(model.state_dict())
    out["points_sha"] = sha(pts)
    path = os.path.join(HERE, f"fullsize_{kind}.npz")
    np.savez_compressed(path, **out)
    print(f"{kind}: oracle {time.time() - t0:.1f} s, {len(dets[0]['scores'])} detections, "
          f"min gap {dets[0]['min_gap']:.3g}, levels "
          f"{[int(v[1].shape[0]) for v in mid['levels'].values()]} -> {path} "
          f"({os.path.getsize(path) / 1024:.0f} KiB)", flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--calibrated"]
    for kind in (args or ["nusc", "nusc1", "waymo", "rcnn"]):
        (make_calibrated if "--calibrated" in sys.argv else make)(kind)
