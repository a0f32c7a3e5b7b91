"""CPU suite: pins the oracle (oracle/) to the reference.  Fixtures under tests/golden/ were
produced by the reference's own code (tests/golden/make_golden.py); when oracle/_ref (the
reference's iou3d_cpu.cpp compiled in place) is present it is also checked live."""
import json
import os

import numpy as np
import pytest

from oracle import detector_oracle as do
from oracle import iou3d, pillar_oracle as po, sparse_oracle as so
from pillarnet_lts_amd import synth

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_iou_known_answer_and_matrix_bit_exact():
    g = load("iou3d.npz")
    assert abs(float(g["known_iou"][0, 0]) - 0.5505213) < 1e-7     # known-answer value of the reference's iou_bev for this pair
    got = iou3d.iou_matrix(g["known_a"], g["known_b"])
    assert got.view(np.uint32)[0, 0] == g["known_iou"].view(np.uint32)[0, 0]
    m = iou3d.iou_matrix(g["boxes_a"], g["boxes_b"])
    assert (m > 0).mean() > 0.02
    np.testing.assert_array_equal(m.view(np.uint32), g["iou_ab"].view(np.uint32))


def test_nms_keep_lists_match_reference():
    g = load("iou3d.npz")
    tags = sorted({k[:-6] for k in g.files if k.endswith("_boxes") and k.startswith("nms_")})
    assert len(tags) == 15
    for t in tags:
        thr = int(t.split("_")[2]) / 100.0
        keep, _ = iou3d.nms(g[t + "_boxes"], thr)
        np.testing.assert_array_equal(keep, g[t + "_keep"])


def test_oracle_vs_compiled_reference_live():
    if iou3d.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    for seed, cluster in [(3, 0.0), (4, 1.0), (5, 3.0)]:
        a = synth.nms_boxes(150, seed=seed, cluster=cluster)
        b = synth.nms_boxes(150, seed=seed + 10, cluster=cluster)
        np.testing.assert_array_equal(iou3d.iou_matrix(a, b).view(np.uint32),
                                      iou3d.ref_iou_matrix(a, b).view(np.uint32))
    deg = np.array([[0, 0, 0, 2, 2, 1, 0], [0, 0, 0, 2, 2, 1, 0], [2, 0, 0, 2, 2, 1, 0],
                    [0, 0, 0, 2, 2, 1, np.pi / 4], [50, 50, 0, 1, 1, 1, 0.1], [0, 0, 0, 0, 0, 0, 0]], np.float32)
    o, r = iou3d.iou_matrix(deg, deg), iou3d.ref_iou_matrix(deg, deg)
    np.testing.assert_array_equal(np.nan_to_num(o, nan=-1).view(np.uint32), np.nan_to_num(r, nan=-1).view(np.uint32))


def test_reader_select_matches_reference_torch_cpu():
    g = load("reader.npz")
    pts = [g["points.0"], g["points.1"]]
    xy, cnt, feats = po.dynamic_pfe_select(pts, [-54, -54, -5.0, 54, 54, 3.0], 0.075, true_divide=True)
    np.testing.assert_array_equal(xy, g["pts_xy"])
    np.testing.assert_array_equal(cnt, g["pts_batch_cnt"])
    np.testing.assert_array_equal(feats, g["pts_features"])
    # the device semantics (reciprocal multiply) differ on a handful of boundary points only
    xy2, cnt2, _ = po.dynamic_pfe_select(pts, [-54, -54, -5.0, 54, 54, 3.0], 0.075, true_divide=False)
    assert abs(int(cnt2.sum()) - int(cnt.sum())) < 50


def test_pfn_mlp_matches_reference_module():
    g = load("reader.npz")
    y = po.pfn_mlp(g["mlp.in"], g["mlp.sd.0.weight"], g["mlp.sd.1.weight"], g["mlp.sd.1.bias"],
                   g["mlp.sd.1.running_mean"], g["mlp.sd.1.running_var"], 1e-3)
    np.testing.assert_allclose(y, g["mlp.out"], rtol=1e-5, atol=1e-5)


def _sd(g, prefix, new_prefix):
    return {new_prefix + k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}


def test_dense_neck_restatement_matches_reference():
    import torch
    g = load("dense_neck.npz")
    sd = _sd(g, "rpnv1.sd.", "neck.")
    y = do.neck_rpnv1(sd, torch.from_numpy(g["rpnv1.in.conv4"]), torch.from_numpy(g["rpnv1.in.conv5"]), (2, 1))
    np.testing.assert_allclose(y[0].numpy(), g["rpnv1.out.0"], rtol=1e-5, atol=1e-5)
    sd = _sd(g, "rpng.sd.", "neck.")
    y = do.neck_rpng(sd, torch.from_numpy(g["rpng.in.conv3"]), torch.from_numpy(g["rpng.in.conv4"]),
                     torch.from_numpy(g["rpng.in.conv5"]), (1, 2))
    np.testing.assert_allclose(y[0].numpy(), g["rpng.out.0"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(y[1].numpy(), g["rpng.out.1"], rtol=1e-5, atol=1e-5)


GOLDEN_TASKS = [dict(stride=8, class_names=["car"]), dict(stride=8, class_names=["bus", "trailer"])]
GOLDEN_HEADS = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)}
GOLDEN_RANGE = [-6, -4.8, -5.0, 6, 4.8, 3.0]


def test_center_head_restatement_matches_reference():
    import torch
    g = load("dense_head.npz")
    sd = _sd(g, "sd.", "bbox_head.")
    preds = do.center_head(sd, (torch.from_numpy(g["in.0"]),), GOLDEN_TASKS, GOLDEN_HEADS)
    for t, pd in enumerate(preds):
        for k, v in pd.items():
            np.testing.assert_allclose(v.numpy(), g[f"out.{t}.{k}"], rtol=1e-5, atol=1e-5)


def test_predict_restatement_matches_reference():
    g, d = load("dense_head.npz"), load("decode.npz")
    with open(os.path.join(G, "decode_test_cfg.json")) as f:
        test_cfg = json.load(f)
    preds = [{k: g[f"out.{t}.{k}"] for k in list(GOLDEN_HEADS) + ["hm"]} for t in range(2)]
    out = do.predict(preds, GOLDEN_TASKS, test_cfg, 0.075, GOLDEN_RANGE)
    for b in range(2):
        assert d[f"{b}.box3d_lidar"].shape[0] > 5
        np.testing.assert_array_equal(out[b]["label_preds"], d[f"{b}.label_preds"])
        np.testing.assert_allclose(out[b]["scores"], d[f"{b}.scores"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(out[b]["box3d_lidar"], d[f"{b}.box3d_lidar"], rtol=1e-5, atol=1e-5)


def test_sparse_gather_form_equals_dense_definition():
    rng = np.random.default_rng(0)
    for (B, H, W, ci, co, n) in [(2, 17, 13, 8, 12, 150), (1, 9, 9, 4, 4, 81), (1, 6, 31, 16, 8, 1)]:
        cells = np.sort(rng.choice(B * H * W, n, replace=False))
        idx = np.stack((cells // (H * W), (cells // W) % H, cells % W), 1).astype(np.int32)
        f = rng.normal(size=(n, ci)).astype(np.float32)
        w = rng.normal(size=(co, 3, 3, ci)).astype(np.float32)
        b = rng.normal(size=co).astype(np.float32)
        o = so.subm_conv2d(f, idx, (H, W), w, b)
        d, _ = so.dense_definition(f, idx, (H, W), B, w, b, True)
        np.testing.assert_allclose(so.to_dense(o, idx, (H, W), B), d, rtol=1e-5, atol=1e-5)
        o2, i2, s2 = so.sparse_conv2d(f, idx, (H, W), B, w, None)
        d2, m2 = so.dense_definition(f, idx, (H, W), B, w, None, False)
        assert i2.shape[0] == int(m2.sum())
        np.testing.assert_allclose(so.to_dense(o2, i2, s2, B), d2, rtol=1e-5, atol=1e-5)


def test_pillar_oracle_invariants():
    pts = [synth.nusc_sweep(5000, seed=1), synth.nusc_sweep(3000, seed=2)]
    rng = np.random.default_rng(0)
    w = rng.normal(0, 0.5, (32, 7)).astype(np.float32)
    args = ([-54, -54, -5.0, 54, 54, 3.0], 0.075, w, np.ones(32), np.zeros(32), np.zeros(32), np.ones(32))
    r = po.pillarize(pts, *args)
    idx = r["indices"].astype(np.int64)
    flat = idx[:, 0] * 1440 * 1440 + idx[:, 1] * 1440 + idx[:, 2]
    assert np.all(np.diff(flat) > 0)
    # permuting the points of a sample changes neither the pillar set nor the pooled features
    perm = [p[np.random.default_rng(5).permutation(len(p))] for p in pts]
    r2 = po.pillarize(perm, *args)
    np.testing.assert_array_equal(r["indices"], r2["indices"])
    np.testing.assert_allclose(r["features"], r2["features"], rtol=0, atol=0)
    assert (r["features"] >= 0).all()


def test_rcnn_second_stage_restatement_matches_reference():
    """BEVStrideFeature / RoIMIXHead / post_process restated in oracle/rcnn_oracle.py vs the
    reference modules' own outputs (tests/golden/rcnn.npz)."""
    from oracle import rcnn_oracle as ro
    g = load("rcnn.npz")
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd.")}
    rng, pillar = g["meta.pc_range"], float(g["meta.pillar"])
    fmap = ro.bev_stride_feature(sd, g["in.bev_feature"], [g["in.conv3"]], p="bev")
    pooled, pts = ro.roi_grid_pool(g["in.rois"], fmap, 7, 4 * pillar, rng[0], rng[1])
    np.testing.assert_allclose(pts, g["out.point_coords"], atol=2e-6)
    B, R = g["in.rois"].shape[:2]
    np.testing.assert_allclose(pooled.reshape(B, R, -1), g["out.roi_features"], atol=2e-4, rtol=1e-4)
    cls, boxes = ro.roi_head(sd, g["out.roi_features"], g["in.rois"], p="roi")
    np.testing.assert_allclose(cls, g["out.batch_cls_preds"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(boxes, g["out.batch_box_preds"], atol=1e-4, rtol=1e-4)
    dets = ro.post_process(g["out.batch_cls_preds"], g["out.batch_box_preds"], g["in.roi_scores"], g["in.roi_labels"])
    for i, d in enumerate(dets):
        np.testing.assert_array_equal(d["label_preds"], g[f"det.{i}.label_preds"])
        np.testing.assert_array_equal(d["box3d_lidar"], g[f"det.{i}.box3d_lidar"])
        np.testing.assert_allclose(d["scores"], g[f"det.{i}.scores"], atol=1e-6)
    assert len(dets[1]["scores"]) == R - 3       # the padded proposals are dropped


def test_waymo_loading_restatement_matches_reference():
    """oracle/loading_oracle.combined_waymo vs the reference's LoadPointCloudFromFile (WaymoDataset branch) on the same
    frames (tests/golden/loading_waymo.npz): bit-exact, one sweep and three."""
    from oracle import loading_oracle as lo
    g = load("loading_waymo.npz")
    frames = [(g[f"xyz.{i}"], g[f"feat.{i}"]) for i in range(3)]
    mats, lags = [None, g["transform.1"], None], [0.0, float(g["time_lag.1"]), float(g["time_lag.2"])]
    np.testing.assert_array_equal(lo.combined_waymo(frames[:1], [None], [0.0]), g["one.points"])
    got = lo.combined_waymo(frames, mats, lags)
    assert got.dtype == np.float32 and got.shape[0] == sum(f[0].shape[0] for f in frames)
    np.testing.assert_array_equal(got, g["three.combined"])


def test_loading_restatement_matches_reference():
    """oracle/loading_oracle.py vs the reference's LoadPointCloudFromFile on the same sweeps (bit-exact: same
    numpy expressions)."""
    from oracle import loading_oracle as lo
    g = load("loading.npz")
    order = [int(i) for i in g["order"]]
    raws = [g["raw.0"]] + [g[f"raw.{i + 1}"] for i in order]
    mats = [None] + [None if np.isnan(g[f"transform.{i}"]).all() else g[f"transform.{i}"] for i in order]
    lags = [0.0] + [float(g[f"time_lag.{i}"]) for i in order]
    got = lo.combined(raws, mats, lags)
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, g["combined"])
    assert got.shape[0] < sum(r.shape[0] for r in raws)      # the close-point filter removed something


# ---- hard voxelizer ("det3d CPU voxelizer") ------------------------------------------------------
@pytest.mark.parametrize("case", ["nusc", "caps", "xyz"])
def test_hard_voxelizer_oracle_matches_reference_points_to_voxel(case):
    """oracle/voxelize_oracle.c and its pure-Python twin against the outputs of the reference's own
    points_to_voxel (det3d/ops/point_cloud/point_cloud_ops.py:112-184), bit-exact: voxel order,
    max_points truncation, max_voxels drop, edge points, both coordinate orders."""
    from oracle import voxelize
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "voxelize.npz"))
    cfg = g[f"{case}.cfg"]
    vs, rng = cfg[:3].tolist(), cfg[3:9].tolist()
    max_points, reverse, max_voxels = int(cfg[9]), bool(cfg[10]), int(cfg[11])
    pts = g[f"{case}.points"]
    for fn in (voxelize.points_to_voxel, voxelize.points_to_voxel_py):
        voxels, coors, num = fn(pts, vs, rng, max_points, reverse, max_voxels)
        np.testing.assert_array_equal(coors, g[f"{case}.coors"])
        np.testing.assert_array_equal(num, g[f"{case}.num"])
        np.testing.assert_array_equal(voxels, g[f"{case}.voxels"])
    assert len(num) > 100


def test_hard_voxelizer_oracle_empty_and_out_of_range():
    from oracle import voxelize
    v, c, n = voxelize.points_to_voxel(np.zeros((0, 5), np.float32), [0.5, 0.5, 8], [-4, -4, -5, 4, 4, 3], 3, True, 10)
    assert v.shape == (0, 3, 5) and c.shape == (0, 3) and n.shape == (0,)
    far = np.full((7, 5), 100.0, np.float32)
    v, c, n = voxelize.points_to_voxel(far, [0.5, 0.5, 8], [-4, -4, -5, 4, 4, 3], 3, True, 10)
    assert v.shape[0] == 0
