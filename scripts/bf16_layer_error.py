"""Relative L2 error of the bf16 inference path against the fp32 path, level by level, on the full-size nuScenes
frame -- for the seeded model as it is (random BatchNorm statistics) and for the calibrated one
(tests/fullsize_cases.apply_calibration)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fullsize_cases import apply_calibration, build_model  # noqa: E402

dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "nusc"
rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-9))
for calibrated in (False, True):
    outs = {}
    for dtype in (torch.float32, torch.bfloat16):
        cfg, model, first_cfg, first, pts = build_model(kind)
        if calibrated:
            apply_calibration(first, np.load(os.path.join(ROOT, "tests", "golden", f"fullsize_{kind}_calibrated.npz")))
        first.to_inference(dtype)
        ex = {"points": [torch.from_numpy(pts).to(dev)], "metadata": [None]}
        o = {}
        with torch.no_grad():
            bev, feats = first.extract_feat(ex)
            for name in ("conv1", "conv2", "conv3", "conv4"):
                f = feats[name]
                o[name] = (f.features[: f.num_active_host()] if hasattr(f, "features") else f).float().clone()
            o["conv5"] = feats["conv5"].float().clone()
            o["bev"] = bev[0].float().clone()
            preds = first.bbox_head(bev)
            o["hm0"] = preds[0]["hm"].float().clone()
            o["reg0"] = preds[0]["reg"].float().clone()
        outs[dtype] = o
        del first, model
    print(f"--- {kind}, {'calibrated' if calibrated else 'random'} BatchNorm statistics")
    for k in outs[torch.float32]:
        a, b = outs[torch.bfloat16][k], outs[torch.float32][k]
        n = min(a.shape[0], b.shape[0])
        centred = rel(a[:n] - b[:n].mean(0, keepdim=True), b[:n] - b[:n].mean(0, keepdim=True)) if a.dim() == 2 else float("nan")
        print(f"{k:10s} rel L2 {rel(a[:n], b[:n]):.5f}   against the fluctuation about the channel mean {centred:.5f}", flush=True)
