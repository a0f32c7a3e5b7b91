"""The seeded full-size cases shared by tests/golden/make_fullsize.py (which runs the CPU oracle on
them once, in the build container) and tests/test_fullsize_gpu.py (which rebuilds the same models
and frames on the GPU box and checks the HIP path against the committed oracle outputs)."""
import hashlib

import numpy as np
import torch

from pillarnet_lts_amd import presets, synth

KINDS = ("nusc", "waymo", "rcnn")


def state_sha(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k].detach().cpu().numpy()).tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def build_model(kind):
    """-> (cfg, model, first-stage model cfg, first-stage detector, points [N,5] fp32).
    nusc: BASELINE configs[1] (PillarNet-18, 1440^2, 10 x 30 000 points); waymo: configs[2]
    (PillarNet-34 + RPNG, 1504^2, 180 000 points); rcnn: configs[4] (Pillar R-CNN on PillarNet-18 +
    RPNG, 1504^2, 180 000 points)."""
    torch.manual_seed(0)
    if kind == "nusc":
        cfg = presets.nusc_pillarnet18()
        model = presets.build(cfg)
        synth.randomize_bn_(model, seed=1)
        first_cfg, first = cfg.model, model
        pts = synth.nusc_frame(10, 30000, seed=0)
    elif kind == "waymo":
        cfg = presets.waymo_pillarnet34_fpn()
        model = presets.build(cfg)
        synth.randomize_bn_(model, seed=1)
        first_cfg, first = cfg.model, model
        pts = synth.waymo_frame(180000, seed=100)
    elif kind == "rcnn":
        cfg = presets.waymo_pillarrcnn_fpn()
        model = presets.build(cfg)
        synth.randomize_bn_(model, seed=1)
        with torch.no_grad():
            model.roi_head.reg_layers[-1].weight.mul_(100.0)
        first_cfg, first = cfg.model.first_stage_cfg, model.single_det
        pts = synth.waymo_frame(180000, seed=300)
    else:
        raise KeyError(kind)
    # spread the heat-map logits: with the init bias alone every score sits at sigmoid(-2.19)
    with torch.no_grad():
        for th in first.bbox_head.task_heads:
            th.hm[-1].weight.mul_(8.0)
    return cfg, model, first_cfg, first, pts
