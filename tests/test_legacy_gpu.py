"""A PillarNet written in the legacy config vocabulary (Identity reader, SpMiddlePillarEncoder18 that pillarises
the points itself, RPNV1 with ds_*/us_* kwargs, CenterHead(in_channels=256, dataset=..., weight=...), geometry in
test_cfg: the shape of configs/nusc/voxelnet/nusc_centerpoint_pillarnet_rpnv1_h.py, on a reduced range) runs on
the same HIP path as the live model: with the live model's weights it returns the same detections (bf16: bit for
bit; fp32: the BEV maps within 1e-4, see below)."""
import logging

import pytest
import torch

import pillarnet_lts_amd as P
from pillarnet_lts_amd import presets, synth

pytestmark = pytest.mark.gpu
RNG = [-12, -12, -5.0, 12, 12, 3.0]


def legacy_cfg():
    tasks = [dict(stride=8, class_names=names) for names in presets.NUSC_TASKS]
    model = dict(
        type="PillarNet", pretrained=None,
        reader=dict(type="Identity", pc_range=RNG, num_input_features=2),
        backbone=dict(type="SpMiddlePillarEncoder18", ds_factor=8, double=2, pc_range=RNG,
                      pillar_cfg=dict(pool0=dict(bev=0.075 / 2), pool1=dict(bev=0.075), pool2=dict(bev=0.15),
                                      pool3=dict(bev=0.3), pool4=dict(bev=0.6))),
        neck=dict(type="RPNV1", layer_nums=[5, 5], ds_layer_strides=[1, 2], ds_num_filters=[256, 256],
                  us_layer_strides=[1, 2], us_num_filters=[128, 128], num_input_features=[256, 256],
                  logger=logging.getLogger("RPN")),
        bbox_head=dict(type="CenterHead", in_channels=256, tasks=tasks, dataset="nuscenes", weight=0.25,
                       code_weights=[1.0] * 6 + [0.2, 0.2, 1.0, 1.0],
                       common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)},
                       share_conv_channel=64, dcn_head=False, iou_reg="GIoU"))
    test_cfg = dict(post_center_limit_range=[-14, -14, -10.0, 14, 14, 10.0], max_per_img=500,
                    nms=dict(use_rotate_nms=True, nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.2),
                    score_threshold=0.1, rectifier=0, pc_range=RNG[:2], out_size_factor=8, voxel_size=[0.075, 0.075])
    return P.ConfigDict(model=model, test_cfg=test_cfg)


def live_cfg():
    cfg = presets.nusc_pillarnet18()
    cfg.model.reader.pc_range = RNG
    cfg.model.bbox_head.point_cloud_range = RNG
    cfg.test_cfg.post_center_limit_range = [-14, -14, -10.0, 14, 14, 10.0]
    return cfg


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_legacy_vocabulary_pillarnet_equals_the_live_model(cuda, dtype):
    torch.manual_seed(0)
    live = presets.build(live_cfg())
    synth.randomize_bn_(live, seed=1)
    with torch.no_grad():
        for th in live.bbox_head.task_heads:
            th.hm[-1].weight.mul_(8.0)
    cfg = legacy_cfg()
    old = P.build_detector(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
    assert old.bbox_head.pillar_size == pytest.approx(0.075) and old.bbox_head.task_strides == [8] * 6
    remap = {}
    for k, v in live.state_dict().items():
        k = k.replace("reader.", "backbone.pfe.", 1) if k.startswith("reader.") else \
            k.replace("backbone.", "backbone.net.", 1) if k.startswith("backbone.") else k
        remap[k] = v
    missing, unexpected = old.load_state_dict(remap, strict=True)
    assert not missing and not unexpected
    live.to_inference(dtype)
    old.to_inference(dtype)
    pts = [torch.from_numpy(synth.nusc_sweep(6000, seed=3, r_max=14.0)).to(cuda),
           torch.from_numpy(synth.nusc_sweep(4500, seed=4, r_max=14.0)).to(cuda)]
    ex = {"points": pts, "metadata": [None, None]}
    with torch.no_grad():
        if dtype == torch.float32:
            # fp32: the dense part runs torch's library convolutions, which do not promise the same summation
            # order for two module instances (seen: 4e-6 on a box coordinate), and a difference of that size can
            # flip an NMS decision -- so the maps are compared, not the detections
            data = dict(points=ex["points"], batch_size=2)
            bev_w, _ = live.extract_feat(data)
            bev_g, _ = old.extract_feat(data)
            for g, w in zip(bev_g, bev_w):
                torch.testing.assert_close(g, w, rtol=1e-4, atol=1e-4)
            return
        want = live(ex, return_loss=False)
        got = old(ex, return_loss=False)
    assert sum(d["scores"].numel() for d in want) > 20
    for g, w in zip(got, want):                      # bf16: every kernel on the path is this library's and deterministic
        for k in ("box3d_lidar", "scores", "label_preds"):
            assert g[k].shape == w[k].shape, k
            assert torch.equal(g[k], w[k]), k


def _legacy_two_stage(cuda, train_cfg=None):

    import numpy as np
    from pillarnet_lts_amd import legacy
    tasks = [dict(num_class=3, class_names=["VEHICLE", "PEDESTRIAN", "CYCLIST"])]
    first = dict(
        type="PointPillars", pretrained=None,
        reader=dict(type="PillarFeatureNet", num_filters=[64, 64], num_input_features=5, with_distance=False,
                    voxel_size=(0.32, 0.32, 6.0), pc_range=(-20.48, -20.48, -2, 20.48, 20.48, 4.0)),
        backbone=dict(type="PointPillarsScatter", ds_factor=1),
        neck=dict(type="RPN", layer_nums=[3, 5, 5], ds_layer_strides=[1, 2, 2], ds_num_filters=[64, 128, 256],
                  us_layer_strides=[1, 2, 4], us_num_filters=[128, 128, 128], num_input_features=64,
                  logger=logging.getLogger("RPN")),
        bbox_head=dict(type="CenterHead", in_channels=128 * 3, tasks=tasks, dataset="waymo", weight=2,
                       code_weights=[1.0] * 8,
                       common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2)}))
    model_cfg = dict(CLASS_AGNOSTIC=True, SHARED_FC=[256, 256], CLS_FC=[256, 256], REG_FC=[256, 256], DP_RATIO=0.3)
    cfg = dict(type="TwoStageDetector", first_stage_cfg=first,
               second_stage_modules=[dict(type="BEVFeatureExtractor", pc_start=[-20.48, -20.48], voxel_size=[0.32, 0.32],
                                          out_stride=1)],
               roi_head=dict(type="RoIHead", input_channels=128 * 3 * 5, model_cfg=model_cfg, code_size=7),
               NMS_POST_MAXSIZE=100, num_point=5, freeze=True)
    test_cfg = P.ConfigDict(post_center_limit_range=[-25, -25, -10.0, 25, 25, 10.0], max_per_img=500,
                            nms=dict(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=500,
                                     nms_post_max_size=100, nms_iou_threshold=0.7),
                            score_threshold=0.1, rectifier=0, pc_range=[-20.48, -20.48], out_size_factor=1,
                            voxel_size=[0.32, 0.32])
    torch.manual_seed(0)
    model = P.build_detector(cfg, train_cfg=train_cfg, test_cfg=test_cfg).to(cuda).eval()
    with torch.no_grad():
        model.single_det.bbox_head.task_heads[0].hm[-1].weight.mul_(8.0)
    step = legacy.Voxelization(cfg=dict(range=[-20.48, -20.48, -2, 20.48, 20.48, 4.0], voxel_size=[0.32, 0.32, 6.0],
                                        max_points_in_voxel=20, max_voxel_num=[8000, 8000]))
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.uniform(-20, 20, (9000, 2)), rng.uniform(-1.5, 3, (9000, 1)), rng.random((9000, 2))],
                         axis=1).astype(np.float32)
    res, _ = step({"mode": "val", "lidar": {"points": pts}}, {})
    vox = res["lidar"]["voxels"]
    batch = P.collate_kitti([dict(metadata={"token": 0}, points=pts, voxels=vox["voxels"], shape=vox["shape"],
                                  num_points=vox["num_points"], num_voxels=vox["num_voxels"],
                                  coordinates=vox["coordinates"])])
    ex = {k: (v.to(cuda) if torch.is_tensor(v) else v) for k, v in batch.items()}
    return model, ex, tasks


def test_legacy_two_stage_pointpillars_runs_end_to_end(cuda):
    """TwoStageDetector(PointPillars first stage, BEVFeatureExtractor with five points per box, RoIHead) in the legacy
    vocabulary (the shape of configs/waymo/pp/two_stage/*.py, small range): voxelise on the host, detect, refine.
    No reference implementation of these types exists, so this checks the contract: every first-stage box comes
    back once, refined, with label, a score in (0, 1] and finite coordinates; bilinear_interpolate_torch against
    a direct evaluation."""
    from pillarnet_lts_amd import legacy
    model, ex, _ = _legacy_two_stage(cuda)
    with torch.no_grad():
        first_boxes, bev, _ = model.single_det.forward_two_stage(dict(ex), return_loss=False)
        out = model(dict(ex), return_loss=False)
    n = first_boxes[0]["box3d_lidar"].shape[0]
    assert 0 < n <= 100 and bev.shape == (1, 384, 128, 128)
    d = out[0]
    assert d["box3d_lidar"].shape == (n, 7) and torch.isfinite(d["box3d_lidar"]).all()
    assert torch.equal(d["label_preds"], first_boxes[0]["label_preds"])
    assert (d["scores"] > 0).all() and (d["scores"] <= 1).all()
    assert float((d["box3d_lidar"] - first_boxes[0]["box3d_lidar"]).abs().max()) > 0          # refined
    # the interpolation helper against a direct evaluation at interior points
    im = torch.rand(7, 9, 3, device=cuda)
    x, y = torch.tensor([1.25, 7.5, 0.0], device=cuda), torch.tensor([2.5, 0.75, 5.0], device=cuda)
    got = legacy.bilinear_interpolate_torch(im, x, y)
    for k in range(3):
        x0, y0 = int(x[k]), int(y[k])
        fx, fy = float(x[k]) - x0, float(y[k]) - y0
        want = (im[y0, x0] * (1 - fx) * (1 - fy) + im[y0 + 1, x0] * (1 - fx) * fy + im[y0, x0 + 1] * fx * (1 - fy) +
                im[y0 + 1, x0 + 1] * fx * fy)
        torch.testing.assert_close(got[k], want, rtol=1e-5, atol=1e-6)


def test_legacy_two_stage_training_branch(cuda):
    """TwoStageDetector.forward(return_loss=True) (two_stage.py:152-195): first-stage loss, RoI targets from the
    first-stage boxes against `gt_boxes_and_cls` (velocity columns dropped for code_size 7), RoIHead class / box
    losses, combine_loss.  The legacy first stage is unpinned, so this checks the contract: the loss dict carries
    the RoI terms, the total is finite, and backward reaches the RoI head (first stage frozen).  ROI_PER_IMAGE equals
    NMS_POST_MAXSIZE: the reference's sampler no longer re-gathers `roi_features` for the sampled RoIs
    (proposal_target_layer.py:79,87-88,117 are commented out), so its branch only runs when the two counts agree."""
    import numpy as np
    from pillarnet_lts_amd import targets
    roi_cfg = dict(TARGET_CONFIG=dict(ROI_PER_IMAGE=100, FG_RATIO=0.5, SAMPLE_ROI_BY_EACH_CLASS=True,
                                      CLS_SCORE_TYPE="roi_iou", CLS_FG_THRESH=0.75, CLS_BG_THRESH=0.25,
                                      CLS_BG_THRESH_LO=0.1, HARD_BG_RATIO=0.8, REG_FG_THRESH=0.55),
                   LOSS_CONFIG=dict(CLS_LOSS="BinaryCrossEntropy", REG_LOSS="L1",
                                    LOSS_WEIGHTS=dict(rcnn_cls_weight=1.0, rcnn_reg_weight=1.0, code_weights=[1.0] * 7)))
    train_cfg = P.ConfigDict(assigner=dict(out_size_factor=1, gaussian_overlap=0.1, max_objs=500, min_radius=2),
                             hm_weight=1, bbox_weight=2)
    model, ex, tasks = _legacy_two_stage(cuda, train_cfg=train_cfg)
    model.roi_head.model_cfg.update(roi_cfg)
    from pillarnet_lts_amd.second_stage import ProposalTargetLayer
    model.roi_head.proposal_target_layer = ProposalTargetLayer(P.ConfigDict(roi_cfg["TARGET_CONFIG"]))
    rng = [-20.48, -20.48, -2, 20.48, 20.48, 4.0]
    rs = np.random.default_rng(3)
    boxes = np.zeros((12, 9), np.float32)
    boxes[:, 0:2] = rs.uniform(-18, 18, (12, 2))
    boxes[:, 2] = rs.normal(0.5, 0.3, 12)
    boxes[:, 3:6] = np.exp(rs.normal(np.log([1.9, 4.5, 1.6]), 0.2, (12, 3)))
    boxes[:, 8] = rs.uniform(-np.pi, np.pi, 12)
    t = targets.assign_labels(boxes, rs.integers(1, 4, 12), [dict(t, stride=1) for t in tasks], rng, 0.32)
    for key in ("hm", "anno_box", "ind", "mask", "cat", "gt_box"):
        ex[key] = [torch.as_tensor(v[None]).to(cuda) for v in t[key]]
    ex["anno_box"] = [a[..., [0, 1, 2, 3, 4, 5, 8, 9]] for a in ex["anno_box"]]      # Waymo heads carry no velocity
    ex["gt_boxes_and_cls"] = torch.as_tensor(t["gt_boxes_and_cls"][None]).to(cuda)
    model.train()
    np.random.seed(1)
    torch.manual_seed(1)
    loss = model(dict(ex), return_loss=True)
    assert "roi_reg_loss" in loss and "roi_cls_loss" in loss and len(loss["roi_cls_loss"]) == len(loss["loss"])
    total = sum(loss["loss"])
    assert torch.isfinite(total)
    total.backward()
    got = [p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.roi_head.parameters()]
    assert all(got) and any(float(p.grad.abs().sum()) > 0 for p in model.roi_head.parameters())
    assert all(p.grad is None for p in model.single_det.parameters())
