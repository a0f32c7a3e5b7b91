"""End-to-end GPU parity of the det3d-API detector (PillarNet-18, nuScenes head) against the CPU
oracle, on a reduced BEV range so the oracle finishes in seconds (all channel widths, strides and
module shapes are the real ones), plus one full-size (1440^2, 300k points) structural check.

Bars (north_star): pillar indices bit-exact; BEV feature maps within 1e-3 (fp32); NMS keep sets
bit-exact given the same head maps.  bf16 (configs[1]): relative L2 of the BEV map <= 3e-2 vs the
fp32 oracle (stated here, the reference has no bf16 mode)."""
import copy

import numpy as np
import pytest
import torch

from oracle import detector_oracle as do
from oracle import sparse_oracle as so
from pillarnet_lts_amd import presets, synth
from pillarnet_lts_amd.center_utils import set_by_task_cfg
from parity_util import assert_detections_match

pytestmark = pytest.mark.gpu


def small_preset():
    cfg = presets.nusc_pillarnet18()
    rng = [-12, -12, -5.0, 12, 12, 3.0]
    cfg.model.reader.pc_range = rng
    cfg.model.bbox_head.point_cloud_range = rng
    cfg.test_cfg.post_center_limit_range = [-14, -14, -10.0, 14, 14, 10.0]
    return cfg


def small_points(seed, n=6000):
    p = synth.nusc_sweep(n, seed=seed, r_max=14.0)
    return p


@pytest.fixture(scope="module")
def built(cuda):
    torch.manual_seed(0)
    cfg = small_preset()
    model = presets.build(cfg)
    synth.randomize_bn_(model, seed=1)
    # spread the heat-map logits so that scores are not all ~sigmoid(-2.19)
    with torch.no_grad():
        for th in model.bbox_head.task_heads:
            th.hm[-1].weight.mul_(8.0)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    test_cfg = set_by_task_cfg(copy.deepcopy(presets.nusc_pillarnet18().test_cfg), model.bbox_head.num_classes)
    test_cfg["post_center_limit_range"] = cfg.test_cfg.post_center_limit_range
    pts = [small_points(3), small_points(4, 4500)]
    dets, mid = do.pillarnet_forward(sd, pts, cfg.model, test_cfg, return_intermediates=True)
    return cfg, model, sd, test_cfg, pts, dets, mid


def _dense_of(level, B):
    return so.to_dense(level[0], level[1], level[2], B)


def test_fp32_stage_by_stage(cuda, built):
    cfg, model, sd, test_cfg, pts, dets, mid = built
    model.to_inference(torch.float32)
    ex = {"points": [torch.from_numpy(p).to(cuda) for p in pts], "metadata": [{}, {}]}
    with torch.no_grad():
        sp = model.reader(ex)
        np.testing.assert_array_equal(sp.indices.cpu().numpy(), mid["reader"]["indices"])
        np.testing.assert_allclose(sp.features.cpu().numpy(), mid["reader"]["features"], rtol=1e-5, atol=1e-5)
        feats = model.backbone(sp)
        for name in ("conv1", "conv2", "conv3"):
            x = feats[name]
            np.testing.assert_array_equal(x.indices.cpu().numpy(), mid["levels"][name][1])
            np.testing.assert_allclose(x.features.cpu().numpy(), mid["levels"][name][0], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(feats["conv4"].cpu().numpy(), mid["conv4"], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(feats["conv5"].cpu().numpy(), mid["conv5"], rtol=1e-3, atol=1e-3)
        bev = model.neck(feats)
        np.testing.assert_allclose(bev[0].cpu().numpy(), mid["bev"][0], rtol=1e-3, atol=1e-3)
        preds = model.bbox_head(bev)
        for t, pd in enumerate(preds):
            for k, v in pd.items():
                np.testing.assert_allclose(v.cpu().numpy(), mid["preds"][t][k], rtol=1e-3, atol=1e-3)


def _as_sets(det):
    b = det["box3d_lidar"]
    return b, det["scores"], det["label_preds"]


def test_decode_and_nms_on_oracle_maps(cuda, built):
    """Same head maps in -> same detections out (order, labels, boxes)."""
    cfg, model, sd, test_cfg, pts, dets, mid = built
    model.to_inference(torch.float32)
    preds = [{k: torch.from_numpy(v).to(cuda) for k, v in d.items()} for d in mid["preds"]]
    out = model.bbox_head.predict({"metadata": [{}, {}]}, preds, model.test_cfg)
    for b in range(2):
        want = dets[b]
        got = {k: out[b][k].cpu().numpy() for k in ("box3d_lidar", "scores", "label_preds")}
        assert_detections_match(got, want, want["min_gap"])
        assert got["box3d_lidar"].shape[0] > 20


def test_fp32_end_to_end_detections(cuda, built):
    cfg, model, sd, test_cfg, pts, dets, mid = built
    model.to_inference(torch.float32)
    ex = {"points": [torch.from_numpy(p).to(cuda) for p in pts], "metadata": [{"token": "a"}, {"token": "b"}]}
    with torch.no_grad():
        out = model(ex, return_loss=False)
    assert out[0]["metadata"] == {"token": "a"}
    for b in range(2):
        want = dets[b]
        got = out[b]["box3d_lidar"].cpu().numpy()
        assert out[b]["label_preds"].dtype == torch.int64
        # maps differ by ~1e-5, so near-tied scores may swap: require that >= 95 % of the oracle's
        # detections have a product detection with the same label within 1 cm / 1e-3 score
        lab = out[b]["label_preds"].cpu().numpy()
        sc = out[b]["scores"].cpu().numpy()
        hit = 0
        for i in range(want["box3d_lidar"].shape[0]):
            d = np.abs(got[:, :2] - want["box3d_lidar"][i, :2]).max(1)
            ok = (d < 1e-2) & (lab == want["label_preds"][i]) & (np.abs(sc - want["scores"][i]) < 1e-3)
            hit += bool(ok.any())
        assert hit >= 0.95 * want["box3d_lidar"].shape[0], (hit, want["box3d_lidar"].shape[0])


# measured on MI355X: 0.869 (hip) / 0.866 (miopen); the bar is that minus a margin.  (Random-init heat maps are nearly
# flat, so ranks flip under bf16 noise: tests/test_fullsize_gpu.py holds the detection-level comparison on a
# well-conditioned model.)
BF16_REDUCED_BAR = 0.84


@pytest.mark.parametrize("backend", ["hip", "miopen"])
def test_bf16_bev_map(cuda, built, backend):
    cfg, model, sd, test_cfg, pts, dets, mid = built
    model.to_inference(torch.bfloat16, dense_backend=backend)
    ex = {"points": [torch.from_numpy(p).to(cuda) for p in pts], "metadata": [{}, {}]}
    with torch.no_grad():
        bev, _ = model.extract_feat(ex)
        got = bev[0].float().cpu().numpy()
    rel = np.linalg.norm(got - mid["bev"][0]) / np.linalg.norm(mid["bev"][0])
    assert rel <= 3e-2, rel
    with torch.no_grad():
        preds = model.bbox_head(bev)
        for t, pd in enumerate(preds):
            for k in ("reg", "height", "dim", "rot", "vel", "hm"):
                g, w = pd[k].float().cpu().numpy(), mid["preds"][t][k]
                assert g.shape == w.shape
                r = np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-6)
                assert r <= 6e-2, (t, k, r)
        out = model(ex, return_loss=False)
    assert out[0]["box3d_lidar"].shape[1] == 9
    # detections: most of the fp32 oracle's boxes survive bf16 (same label, centre within 5 cm)
    want = dets[0]
    got_b, lab = out[0]["box3d_lidar"].cpu().numpy(), out[0]["label_preds"].cpu().numpy()
    hit = sum(bool(((np.abs(got_b[:, :2] - want["box3d_lidar"][i, :2]).max(1) < 5e-2) &
                    (lab == want["label_preds"][i])).any()) for i in range(want["box3d_lidar"].shape[0]))
    n = want["box3d_lidar"].shape[0]
    print(f"\n[{backend}] reduced-range bf16 detections with a partner (label, centre 5 cm): {hit}/{n} = {hit / n:.3f}")
    assert hit >= BF16_REDUCED_BAR * n, (hit, n)
    model.to_inference(torch.float32)


def test_full_size_frame_structure(cuda):
    """configs[1] shapes (1440^2 grid, 300k points): output contract and invariants."""
    torch.manual_seed(0)
    cfg = presets.nusc_pillarnet18()
    model = presets.build(cfg)
    synth.randomize_bn_(model, seed=1)
    model.to_inference(torch.bfloat16)
    pts = torch.from_numpy(synth.nusc_frame(10, 30000, seed=0)).to(cuda)
    with torch.no_grad():
        out = model({"points": [pts], "metadata": [None]}, return_loss=False)
        out2 = model({"points": [pts], "metadata": [None]}, return_loss=False)
    d = out[0]
    n = d["box3d_lidar"].shape[0]
    assert d["box3d_lidar"].shape == (n, 9) and d["scores"].shape == (n,) and d["label_preds"].shape == (n,)
    assert n <= 6 * 83
    assert (d["scores"] > 0.1).all() and (d["label_preds"] >= 0).all() and (d["label_preds"] < 10).all()
    # the whole HIP path is bit-reproducible (no float atomics anywhere): same frame twice ->
    # identical detections.  (The MIOpen dense backend is not: its strided bf16 conv differs by a
    # bf16 ulp between runs.)
    assert torch.equal(d["box3d_lidar"], out2[0]["box3d_lidar"]) and torch.equal(d["scores"], out2[0]["scores"])
    # per task: scores are non-increasing inside each task's block of the concatenation
    lab = d["label_preds"].cpu().numpy()
    sc = d["scores"].cpu().numpy()
    task_of = np.array([0, 1, 1, 2, 2, 3, 4, 4, 5, 5])[lab]
    assert np.all(np.diff(task_of) >= 0)
    for t in range(6):
        s = sc[task_of == t]
        assert np.all(np.diff(s) <= 1e-7)


# ---- configs[2]: PillarNet-34 + RPNG two-scale neck + per-class NMS + IoU head (Waymo preset) -------
def small_waymo():
    cfg = presets.waymo_pillarnet34_fpn()
    rng = [-12.8, -12.8, -2, 12.8, 12.8, 4.0]
    cfg.model.reader.pc_range = rng
    cfg.model.bbox_head.point_cloud_range = rng
    cfg.test_cfg.post_center_limit_range = [-15, -15, -10.0, 15, 15, 10.0]
    return cfg


@pytest.fixture(scope="module")
def built_waymo(cuda):
    torch.manual_seed(1)
    cfg = small_waymo()
    model = presets.build(cfg)
    synth.randomize_bn_(model, seed=2)
    with torch.no_grad():
        for th in model.bbox_head.task_heads:
            th.hm[-1].weight.mul_(8.0)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    pts = [synth.waymo_frame(7000, seed=100)[:, :5] * np.array([0.17, 0.17, 1, 1, 1], np.float32)]
    dets, mid = do.pillarnet_forward(sd, pts, cfg.model, model.test_cfg, return_intermediates=True)
    return cfg, model, pts, dets, mid


def test_waymo_fp32_maps_and_detections(cuda, built_waymo):
    cfg, model, pts, dets, mid = built_waymo
    model.to_inference(torch.float32)
    ex = {"points": [torch.from_numpy(p).to(cuda) for p in pts], "metadata": [None]}
    with torch.no_grad():
        bev, feats = model.extract_feat(ex)
        assert len(bev) == 2
        for k in range(2):
            np.testing.assert_allclose(bev[k].cpu().numpy(), mid["bev"][k], rtol=1e-3, atol=1e-3)
        # decode + per-class NMS on the oracle's maps reproduces the oracle's detections
        preds = [{k: torch.from_numpy(v).to(cuda) for k, v in d.items()} for d in mid["preds"]]
        out = model.bbox_head.predict({"metadata": [None]}, preds, model.test_cfg)
    want = dets[0]
    assert out[0]["box3d_lidar"].shape[1] == 7
    got = {k: out[0][k].cpu().numpy() for k in ("box3d_lidar", "scores", "label_preds")}
    assert_detections_match(got, want, want["min_gap"])
    assert want["box3d_lidar"].shape[0] > 10


def test_waymo_bf16_hip(cuda, built_waymo):
    cfg, model, pts, dets, mid = built_waymo
    model.to_inference(torch.bfloat16)
    ex = {"points": [torch.from_numpy(p).to(cuda) for p in pts], "metadata": [None]}
    with torch.no_grad():
        bev, _ = model.extract_feat(ex)
        for k in range(2):
            got = bev[k].float().cpu().numpy()
            rel = np.linalg.norm(got - mid["bev"][k]) / np.linalg.norm(mid["bev"][k])
            assert rel <= 4e-2, (k, rel)
        out = model(ex, return_loss=False)
    assert out[0]["box3d_lidar"].shape[1] == 7 and out[0]["label_preds"].max() <= 2
    model.to_inference(torch.float32)


def test_degenerate_frames(cuda):
    """Empty / out-of-range / single-point frames run the whole HIP path (row counts of 0 reach every kernel)."""
    torch.manual_seed(0)
    cfg = small_preset()
    model = presets.build(cfg)
    synth.randomize_bn_(model, seed=1)
    for dtype in (torch.bfloat16, torch.float32):
        model.to_inference(dtype)
        frames = [np.zeros((0, 5), np.float32), np.full((50, 5), 400.0, np.float32),
                  np.array([[0.1, 0.2, -1.0, 3.0, 0.0]], np.float32)]
        with torch.no_grad():
            out = model({"points": [torch.from_numpy(f).to(cuda) for f in frames], "metadata": [None] * 3},
                        return_loss=False)
        assert len(out) == 3
        # an empty BEV map gives the same (bias-only) head output: frames 0 and 1 agree (exactly on the HIP
        # dense path; MIOpen's fp32 algorithms may differ in the last bits between batch slots)
        if dtype == torch.bfloat16:
            assert torch.equal(out[0]["scores"], out[1]["scores"])
        else:
            # MIOpen's fp32 convolutions differ in the last bits between batch slots and between runs: the
            # bias-only maps are full of near-ties, so the two empty frames may keep 432 vs 433 boxes
            assert abs(out[0]["scores"].shape[0] - out[1]["scores"].shape[0]) <= 4
            torch.testing.assert_close(out[0]["scores"][:50], out[1]["scores"][:50], rtol=1e-4, atol=1e-5)
        for d in out:
            assert d["box3d_lidar"].shape[1] == 9 and torch.isfinite(d["box3d_lidar"]).all()


def test_waymo_full_size_structure(cuda):
    """configs[2] at full size: PillarNet-34, 0.1 m pillars (1504^2 grid), 180k points, RPNG, 3 classes with
    per-class NMS (pre 2048/1024/1024, post 200/150/150) and the IoU head, bf16 HIP path."""
    torch.manual_seed(0)
    cfg = presets.waymo_pillarnet34_fpn()
    model = presets.build(cfg)
    synth.randomize_bn_(model, seed=1)
    model.to_inference(torch.bfloat16)
    pts = torch.from_numpy(synth.waymo_frame(180000, seed=100)).to(cuda)
    with torch.no_grad():
        out = model({"points": [pts], "metadata": [None]}, return_loss=False)
        out2 = model({"points": [pts], "metadata": [None]}, return_loss=False)
    d = out[0]
    n = d["box3d_lidar"].shape[0]
    assert d["box3d_lidar"].shape == (n, 7) and n <= 500 and n > 0
    assert int(d["label_preds"].max()) <= 2 and (d["scores"] > 0.1).all()
    lab = d["label_preds"].cpu().numpy()
    assert np.all(np.diff(lab) >= 0)          # VEHICLE block, then PEDESTRIAN, then CYCLIST
    for k, cap in enumerate((200, 150, 150)):
        assert int((lab == k).sum()) <= cap
    assert torch.equal(d["box3d_lidar"], out2[0]["box3d_lidar"])
