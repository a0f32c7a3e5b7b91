"""Prints how many of the fp32 oracle's full-size detections the bf16 product path reproduces (same label, centre
<= 5 cm, score <= 2e-2) on the calibrated models (tests/golden/fullsize_*_calibrated.npz), and for the detections
without a partner what separates them from the nearest product detection of their label."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fullsize_cases import apply_calibration, build_model  # noqa: E402
from parity_util import partner_rate  # noqa: E402

dev = torch.device("cuda:0")
for kind in sys.argv[1:] or ["nusc", "nusc1", "waymo", "rcnn"]:
    cfg, model, first_cfg, first, pts = build_model(kind)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"fullsize_{kind}_calibrated.npz"))
    apply_calibration(first, g)
    want = dict(box3d_lidar=g["det_boxes"], scores=g["det_scores"], label_preds=g["det_labels"])
    ex = {"points": [torch.from_numpy(pts).to(dev)], "metadata": [None]}
    for dtype in (torch.float32, torch.bfloat16):
        first.to_inference(dtype)
        with torch.no_grad():
            out = first(ex, return_loss=False)[0]
        got = {k: out[k].float().cpu().numpy() if k != "label_preds" else out[k].cpu().numpy()
               for k in ("box3d_lidar", "scores", "label_preds")}
        for ct, st in ((1e-2, 1e-3), (5e-2, 2e-2), (1e-1, 5e-2), (1e-1, 1.0)):
            r, h, n = partner_rate(got, want, ct, st)
            print(f"{kind} {str(dtype)[6:]}: centre {ct} score {st}: {h}/{n} = {r:.3f}  (product returned {got['scores'].shape[0]})",
                  flush=True)
        if dtype == torch.bfloat16:
            dc, ds, gone = [], [], 0
            for i in range(want["scores"].shape[0]):
                same = np.flatnonzero(got["label_preds"] == want["label_preds"][i])
                d = np.abs(got["box3d_lidar"][same, :2] - want["box3d_lidar"][i, :2]).max(1)
                j = same[d.argmin()]
                if d.min() > 0.3:
                    gone += 1
                    continue
                dc.append(d.min())
                ds.append(abs(got["scores"][j] - want["scores"][i]))
            print(f"   nearest same-label product box: absent (> 0.3 m) for {gone}; centre error quantiles [m] "
                  f"{np.quantile(dc, [.5, .9, .99, 1]).round(4)}; score error quantiles {np.quantile(ds, [.5, .9, .99, 1]).round(4)}; "
                  f"box error quantiles (all 9) {np.quantile(np.abs(got['box3d_lidar'][:1] * 0), [0]).round(3)}", flush=True)
