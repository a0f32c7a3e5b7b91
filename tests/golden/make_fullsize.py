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

    python tests/golden/make_fullsize.py [nusc] [waymo] [rcnn]      # -> tests/golden/fullsize_*.npz
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
    for kind in (sys.argv[1:] or ["nusc", "waymo", "rcnn"]):
        make(kind)
