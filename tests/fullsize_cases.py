"""The seeded full-size cases shared by tests/golden/make_fullsize.py (which runs the CPU oracle on
them once, in the build container) and tests/test_fullsize_gpu.py (which rebuilds the same models
and frames on the GPU box and checks the HIP path against the committed oracle outputs)."""
import hashlib

import numpy as np
import torch

from pillarnet_lts_amd import presets, synth

KINDS = ("nusc", "nusc1", "waymo", "rcnn")


def state_sha(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k].detach().cpu().numpy()).tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def apply_calibration(first, g):
    """Turns the seeded random model into one that behaves numerically like a trained network, from a committed fixture
    `g` (tests/golden/fullsize_<kind>_calibrated.npz, written by tests/golden/make_fullsize.py --calibrated; its `set.<key>`
    arrays replace the state-dict entries of that name):

    * every backbone / neck / head BatchNorm gets the running statistics the oracle measured on this very frame, and
      bias = 1.5 x weight.  With random statistics the maps are a constant plus a fluctuation of a few per cent, and
      which 1000 cells clear the pre-NMS cut is decided in the fourth decimal.  With calibrated statistics and the bias
      at zero a random-weight BatchNorm-ReLU stack is chaotic: any perturbation grows about 1.2x per layer (measured
      on the nuScenes frame: bf16 against fp32 0.7 % after the first stage, 15.6 % at the BEV map), which a trained
      network does not do.  1.5 standard deviations above zero, 93 % of the units pass the ReLU and the growth
      is gone (0.3 % -> 1.0 %);
    * the output convs of the head are mean-centred per output channel and scaled so that the regression outputs have
      O(1) statistics (offsets inside a cell, log sizes, ...) and only the top 1 % of the heat map clears the 0.1
      score threshold, the scores above it spread over about four logits -- as a trained head's do.
    Nothing a detection comparison between bf16 and fp32 can rest on exists without these two steps."""
    sd = first.state_dict()
    with torch.no_grad():
        for key in g.files:
            if key.startswith("set."):
                sd[key[4:]].copy_(torch.from_numpy(g[key]))


def build_model(kind):
    """-> (cfg, model, first-stage model cfg, first-stage detector, points [N,5] fp32).
    nusc: BASELINE configs[1] (PillarNet-18, 1440^2, 10 x 30 000 points); nusc1: configs[0]'s frame (the same model,
    ONE sweep of 30 000 points); waymo: configs[2]
    (PillarNet-34 + RPNG, 1504^2, 180 000 points); rcnn: configs[4] (Pillar R-CNN on PillarNet-18 +
    RPNG, 1504^2, 180 000 points)."""
    torch.manual_seed(0)
    if kind in ("nusc", "nusc1"):
        cfg = presets.nusc_pillarnet18()
        model = presets.build(cfg)
        synth.randomize_bn_(model, seed=1)
        first_cfg, first = cfg.model, model
        pts = synth.nusc_frame(10 if kind == "nusc" else 1, 30000, seed=0)
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
        for t, th in enumerate(first.bbox_head.task_heads):
            th.hm[-1].weight.mul_(8.0)
    return cfg, model, first_cfg, first, pts
