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
