"""The lazily evaluated CenterHead (heat maps densely, regression towers + output convs only at the cells
predict() ranks) against the dense HIP head: identical detections -- boxes, scores and labels bit for bit,
because every value comes out of the same kernels in the same summation order -- on the reduced-range
nuScenes and Waymo (per-class NMS, two feature levels) models, at full size, and through the fallback that
redoes a frame densely when the centre-range mask would have changed the ranking.
Reference semantics: det3d/models/bbox_heads/center_head.py:14-51,116-127,217-413."""
import os

import numpy as np
import pytest
import torch

from pillarnet_lts_amd import presets, synth
from test_detector_gpu import small_points, small_preset, small_waymo

pytestmark = pytest.mark.gpu


def _both(model, ex):
    """(lazy, dense) detections of the same frame; asserts the lazy path really ran."""
    assert model.bbox_head.lazy_plan(model.test_cfg) is not None
    os.environ.pop("PN_HEAD_DENSE", None)
    with torch.no_grad():
        lazy = model(ex, return_loss=False)
    os.environ["PN_HEAD_DENSE"] = "1"
    try:
        assert model.bbox_head.lazy_plan(model.test_cfg) is None
        with torch.no_grad():
            dense = model(ex, return_loss=False)
    finally:
        os.environ.pop("PN_HEAD_DENSE", None)
    return lazy, dense


def _same(lazy, dense, min_boxes):
    assert len(lazy) == len(dense)
    for a, b in zip(lazy, dense):
        assert a["box3d_lidar"].shape == b["box3d_lidar"].shape
        assert torch.equal(a["label_preds"], b["label_preds"])
        assert torch.equal(a["scores"], b["scores"])
        assert torch.equal(a["box3d_lidar"], b["box3d_lidar"])
        assert a["metadata"] == b["metadata"]
    assert sum(a["box3d_lidar"].shape[0] for a in lazy) >= min_boxes


def _nusc_small(cuda):
    torch.manual_seed(0)
    cfg = small_preset()
    model = presets.build(cfg)
    synth.randomize_bn_(model, seed=1)
    with torch.no_grad():
        for th in model.bbox_head.task_heads:
            th.hm[-1].weight.mul_(8.0)
    model.to_inference(torch.bfloat16)
    pts = [small_points(3), small_points(4, 4500)]
    ex = {"points": [torch.from_numpy(p).to(cuda) for p in pts], "metadata": [{"token": "a"}, {"token": "b"}]}
    return model, ex


def test_lazy_equals_dense_nusc_batch2(cuda):
    model, ex = _nusc_small(cuda)
    _same(*_both(model, ex), min_boxes=40)


def test_lazy_equals_dense_waymo_two_levels_per_class_nms(cuda):
    torch.manual_seed(1)
    cfg = small_waymo()
    cfg.model.bbox_head.common_heads = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2)}
    model = presets.build(cfg)
    synth.randomize_bn_(model, seed=2)
    with torch.no_grad():
        for th in model.bbox_head.task_heads:
            th.hm[-1].weight.mul_(8.0)
    model.to_inference(torch.bfloat16)
    pts = synth.waymo_frame(7000, seed=100)[:, :5] * np.array([0.17, 0.17, 1, 1, 1], np.float32)
    ex = {"points": [torch.from_numpy(pts).to(cuda)], "metadata": [None]}
    _same(*_both(model, ex), min_boxes=10)


def test_rectified_scores_and_double_flip_keep_the_dense_head(cuda):
    torch.manual_seed(1)
    model = presets.build(small_waymo())          # has the IoU head
    model.to_inference(torch.bfloat16)
    assert model.bbox_head.lazy_plan(model.test_cfg) is not None      # rectifier 0: IoU map not needed for ranking
    cfg = dict(model.test_cfg, rectifier=[[0.5], [0.5, 0.5]])
    assert model.bbox_head.lazy_plan(cfg) is None
    assert model.bbox_head.lazy_plan(dict(model.test_cfg, double_flip=True)) is None
    model.to_inference(torch.float32)
    assert model.bbox_head.lazy_plan(model.test_cfg) is None


def test_range_mask_fallback_redoes_the_frame_densely(cuda):
    """A limit range that cuts through the ranked cells: the lazy path cannot know which cells to rank instead
    (the reference masks before it sorts), reports it, and the frame is redone by the dense head."""
    model, ex = _nusc_small(cuda)
    model.test_cfg["post_center_limit_range"] = [-4.0, -4.0, -10.0, 4.0, 4.0, 10.0]
    lazy, dense = _both(model, ex)
    _same(lazy, dense, min_boxes=5)
    ctr = torch.cat([d["box3d_lidar"][:, :2] for d in lazy])
    assert float(ctr.abs().max()) <= 4.0


def test_lazy_equals_dense_full_size(cuda):
    """configs[1]: 1440^2 grid, 300 000 points, 6 tasks x 1000 ranked cells."""
    torch.manual_seed(0)
    model = presets.build(presets.nusc_pillarnet18())
    synth.randomize_bn_(model, seed=1)
    model.to_inference(torch.bfloat16)
    pts = torch.from_numpy(synth.nusc_frame(10, 30000, seed=0)).to(cuda)
    _same(*_both(model, {"points": [pts], "metadata": [None]}), min_boxes=100)
