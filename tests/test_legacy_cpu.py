"""Legacy config vocabulary (configs/nusc/**, configs/waymo/**: CenterPoint-era type names and kwargs the
reference's own registry no longer resolves).  Construction and the host voxelizer are covered here; the
numerics of these adapters have no counterpart in the reference to pin against ("parity unpinned"), except
the hard voxelizer, which is pinned to the reference's points_to_voxel outputs in tests/golden/voxelize.npz."""
import glob
import logging
import os

import numpy as np
import pytest
import torch

import pillarnet_lts_amd as P
from pillarnet_lts_amd import legacy

REF = "/root/reference/configs"
G = os.path.join(os.path.dirname(__file__), "golden")


def legacy_files():
    return sorted(glob.glob(f"{REF}/nusc/**/*.py", recursive=True) + glob.glob(f"{REF}/waymo/**/*.py", recursive=True))


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's config files are only in the build container")
def test_every_legacy_config_file_parses_and_builds():
    files = legacy_files()
    assert len(files) == 30
    kinds = set()
    for f in files:
        cfg = P.Config.fromfile(f)
        model = P.build_detector(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.test_cfg)
        assert sum(p.numel() for p in model.parameters()) > 1e5, f
        kinds.add(type(model).__name__)
        first = getattr(model, "single_det", model)
        assert all(s > 0 for s in first.bbox_head.task_strides), f      # strides resolved from test_cfg
    assert kinds == {"PointPillars", "PillarNet", "VoxelNet", "TwoStageDetector", "PillarRCNN"}


@pytest.mark.parametrize("case", ["nusc", "caps", "xyz"])
def test_library_points_to_voxel_matches_reference_outputs(case):
    """the library's host voxelizer (pn_points_to_voxel) against the reference's points_to_voxel
    (det3d/ops/point_cloud/point_cloud_ops.py:112-184) outputs, bit-exact."""
    g = np.load(os.path.join(G, "voxelize.npz"))
    cfg = g[f"{case}.cfg"]
    voxels, coors, num = legacy.points_to_voxel(g[f"{case}.points"], cfg[:3].tolist(), cfg[3:9].tolist(), int(cfg[9]),
                                                bool(cfg[10]), int(cfg[11]))
    np.testing.assert_array_equal(coors, g[f"{case}.coors"])
    np.testing.assert_array_equal(num, g[f"{case}.num"])
    np.testing.assert_array_equal(voxels, g[f"{case}.voxels"])


def test_points_to_voxel_empty_and_out_of_range():
    v, c, n = legacy.points_to_voxel(np.zeros((0, 5), np.float32), [0.5, 0.5, 8], [-4, -4, -5, 4, 4, 3], 3, True, 10)
    assert v.shape == (0, 3, 5) and c.shape == (0, 3) and n.shape == (0,)
    v, c, n = legacy.points_to_voxel(np.full((7, 5), 100.0, np.float32), [0.5, 0.5, 8], [-4, -4, -5, 4, 4, 3], 3, True, 10)
    assert v.shape[0] == 0


def pp_config():
    """a PointPillars model in the legacy vocabulary (the shape of configs/nusc/pp/*.py on a small range)."""
    tasks = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "bus"])]
    return dict(
        type="PointPillars", pretrained=None,
        reader=dict(type="PillarFeatureNet", num_filters=[64, 64], num_input_features=5, with_distance=False,
                    voxel_size=(0.2, 0.2, 8), pc_range=(-12.8, -12.8, -5.0, 12.8, 12.8, 3.0)),
        backbone=dict(type="PointPillarsScatter", ds_factor=1),
        neck=dict(type="RPN", layer_nums=[3, 5, 5], ds_layer_strides=[2, 2, 2], ds_num_filters=[64, 128, 256],
                  us_layer_strides=[0.5, 1, 2], us_num_filters=[128, 128, 128], num_input_features=64,
                  logger=logging.getLogger("RPN")),
        bbox_head=dict(type="CenterHead", in_channels=384, tasks=tasks, dataset="nuscenes", weight=0.25,
                       code_weights=[1.0] * 8 + [0.2, 0.2],
                       common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)}),
    )


def test_pointpillars_legacy_model_runs_from_the_voxelization_step():
    torch.manual_seed(0)
    test_cfg = dict(post_center_limit_range=[-15, -15, -10, 15, 15, 10], max_per_img=100,
                    nms=dict(nms_pre_max_size=100, nms_post_max_size=20, nms_iou_threshold=0.2), score_threshold=0.1,
                    pc_range=[-12.8, -12.8], out_size_factor=4, voxel_size=[0.2, 0.2])
    model = P.build_detector(pp_config(), train_cfg=None, test_cfg=test_cfg).eval()
    assert model.bbox_head.task_strides == [4, 4] and model.bbox_head.pillar_size == pytest.approx(0.2)
    step = legacy.Voxelization(cfg=dict(range=[-12.8, -12.8, -5.0, 12.8, 12.8, 3.0], voxel_size=[0.2, 0.2, 8],
                                        max_points_in_voxel=20, max_voxel_num=[3000, 6000]))
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.uniform(-12.8, 12.8, (4000, 2)), rng.uniform(-4, 2, (4000, 1)), rng.random((4000, 2))],
                         axis=1).astype(np.float32)
    res, _ = step({"mode": "val", "lidar": {"points": pts}}, {})
    vox = res["lidar"]["voxels"]
    assert vox["voxels"].shape[1:] == (20, 5) and vox["shape"].tolist() == [128, 128, 1]
    batch = P.collate_kitti([dict(metadata={"token": 0}, points=pts, voxels=vox["voxels"], shape=vox["shape"],
                                  num_points=vox["num_points"], num_voxels=vox["num_voxels"],
                                  coordinates=vox["coordinates"])])
    with torch.no_grad():
        data = dict(features=batch["voxels"], num_voxels=batch["num_points"], coors=batch["coordinates"],
                    batch_size=1, input_shape=batch["shape"][0])
        x = model.extract_feat(data)
        x = torch.stack(x) if isinstance(x, tuple) else x
        assert x.shape == (1, 384, 32, 32)
        preds = model.bbox_head((x,))
    assert len(preds) == 2 and preds[0]["hm"].shape == (1, 1, 32, 32) and preds[1]["hm"].shape == (1, 2, 32, 32)
    assert all(torch.isfinite(v).all() for pd in preds for v in pd.values())


def test_voxelnet_backbone_says_what_it_lacks():
    net = legacy.SpMiddleResNetFHD(num_input_features=5)
    with pytest.raises(NotImplementedError, match="3-D sparse convolution"):
        net(None, None, 1, None)
