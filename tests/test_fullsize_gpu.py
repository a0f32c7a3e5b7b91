"""Full-size parity (BASELINE configs[1], [2], [4]; configs[3] is tests/test_train_gpu.py's
full-size step): the HIP path at the real grid / point counts against oracle outputs committed as
compact fixtures (tests/golden/fullsize_*.npz, produced by tests/golden/make_fullsize.py from the
same seeds; the oracle needs 30-70 s per frame, so it ran once in the build container).

Bars: row counts, (b, y, x) rows (sha256 of the whole int32 array) and occupancy bitmaps bit-exact;
fp32 features / maps 1e-3 abs + rel at the sampled rows / cells; decode + NMS from the product's own
fp32 head maps against the oracle's predict run on the host: labels exact, scores 2e-6, boxes 1e-5,
borderline IoUs bounded (tests/parity_util.py); bf16: relative L2 over the sampled BEV cells <= 3e-2
(4e-2 for the two-scale Waymo neck), head maps <= 6e-2 (the reference has no bf16 mode)."""
import copy
import os

import numpy as np
import pytest
import torch

from fullsize_cases import KINDS, apply_calibration, build_model, sha, state_sha
from oracle import detector_oracle as do
from parity_util import assert_detections_match, partner_rate

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
_cache = {}


def case(kind):
    if kind not in _cache:
        _cache.clear()                           # one full-size model at a time
        cfg, model, first_cfg, first, pts = build_model(kind)
        _cache[kind] = (cfg, model, first_cfg, first, pts, np.load(os.path.join(G, f"fullsize_{kind}.npz")))
    return _cache[kind]


def close(got, want, what, tol=1e-3):
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol, err_msg=what)


def cells_of(t, cells):
    """[1, C, H, W] device tensor -> [len(cells), C] numpy at flat cells."""
    C = t.shape[1]
    return t[0].reshape(C, -1)[:, torch.from_numpy(cells).to(t.device)].T.float().cpu().numpy()


@pytest.mark.parametrize("kind", KINDS)
def test_fixture_describes_this_model_and_frame(kind):
    cfg, model, first_cfg, first, pts, g = case(kind)
    assert np.array_equal(sha(pts), g["points_sha"]), "synthetic frame differs from the one the oracle saw"
    assert np.array_equal(state_sha(model.state_dict()), g["weights_sha"]), \
        "seeded weights differ from the ones the oracle saw (regenerate with tests/golden/make_fullsize.py)"


@pytest.mark.parametrize("kind", KINDS)
def test_fp32_full_size_matches_oracle_fixture(cuda, kind):
    cfg, model, first_cfg, first, pts, g = case(kind)
    first.to_inference(torch.float32)
    ex = {"points": [torch.from_numpy(pts).to(cuda)], "metadata": [None]}
    with torch.no_grad():
        sp = first.reader(ex)
        idx = sp.indices.cpu().numpy()
        assert idx.shape[0] == int(g["reader_count"])
        assert np.array_equal(sha(idx.astype(np.int32)), g["reader_idx_sha"]), "pillar (b,y,x) rows differ"
        rows = g["reader_rows"]
        np.testing.assert_array_equal(idx[rows], g["reader_idx_rows"])
        close(sp.features[torch.from_numpy(rows).to(cuda)].cpu().numpy(), g["reader_feat"], "reader", 1e-5)
        x = sp
        for stage in (1, 2, 3, 4):
            name = f"conv{stage}"
            x = getattr(first.backbone, name)(x)
            idx = x.indices.cpu().numpy()
            assert idx.shape[0] == int(g[f"{name}_count"]), name
            assert np.array_equal(sha(idx.astype(np.int32)), g[f"{name}_idx_sha"]), f"{name}: active rows differ"
            if stage > 1:
                H, W = (int(v) for v in g[f"{name}_shape"])
                occ = np.zeros((H, W), bool)
                occ[idx[:, 1], idx[:, 2]] = True
                np.testing.assert_array_equal(np.packbits(occ), g[f"{name}_occ"])
            rows = g[f"{name}_rows"]
            close(x.features[torch.from_numpy(rows).to(cuda)].cpu().numpy(), g[f"{name}_feat"], name)
        feats = first.backbone(sp)
        close(cells_of(feats["conv4"], g["dense4_cells"]), g["dense4_vals"], "conv4.dense()")
        close(cells_of(feats["conv5"], g["dense5_cells"]), g["dense5_vals"], "conv5")
        bev = first.neck(feats)
        for k in range(len(bev)):
            close(cells_of(bev[k], g[f"bev{k}_cells"]), g[f"bev{k}_vals"], f"bev{k}")
        preds = first.bbox_head(bev)
        for t, pd in enumerate(preds):
            for name, v in pd.items():
                close(cells_of(v, g[f"pred{t}_{name}_cells"]), g[f"pred{t}_{name}_vals"], f"task {t} {name}")

        # decode + NMS at full size: the oracle's predict (host, numpy + the C NMS oracle) on the
        # product's own fp32 head maps against the product's detections from the same maps
        out = first.bbox_head.predict({"metadata": [None]}, preds, first.test_cfg)
    preds_np = [{k: v.float().cpu().numpy() for k, v in pd.items()} for pd in preds]
    # scores: torch's own sigmoid on the device, which is what the reference calls (center_head.py:230); the
    # host's exp differs from the device's in the last bit on about a quarter of the cells, which reorders the
    # near-tied scores a random-init head is full of
    dev_sigmoid = lambda x: torch.sigmoid(torch.from_numpy(np.ascontiguousarray(x)).to(cuda)).cpu().numpy()
    want = do.predict(preds_np, first_cfg["bbox_head"]["tasks"], copy.deepcopy(first.test_cfg),
                      first_cfg["reader"]["pillar_size"], first_cfg["reader"]["pc_range"], sigmoid=dev_sigmoid)[0]
    got = {k: out[0][k].cpu().numpy() for k in ("box3d_lidar", "scores", "label_preds")}
    assert want["box3d_lidar"].shape[0] >= 100
    assert_detections_match(got, want, want["min_gap"])

    # and against the oracle's own end-to-end detections (its maps, the host's sigmoid): maps agree to ~1e-5
    # and a random-init head's scores sit within 1e-3 of each other, so near ties reorder at the pre_max /
    # post_max cuts -- at least 85 % of the oracle's boxes have a partner (same label, centre 1 cm, score 1e-3)
    hit = 0
    for i in range(g["det_boxes"].shape[0]):
        ok = (np.abs(got["box3d_lidar"][:, :2] - g["det_boxes"][i, :2]).max(1) < 1e-2) & \
             (got["label_preds"] == g["det_labels"][i]) & (np.abs(got["scores"] - g["det_scores"][i]) < 1e-3)
        hit += bool(ok.any())
    assert hit >= 0.85 * g["det_boxes"].shape[0], (hit, g["det_boxes"].shape[0])


@pytest.mark.parametrize("kind", KINDS)
def test_bf16_full_size_within_stated_tolerance(cuda, kind):
    cfg, model, first_cfg, first, pts, g = case(kind)
    first.to_inference(torch.bfloat16)
    ex = {"points": [torch.from_numpy(pts).to(cuda)], "metadata": [None]}
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-6))
    with torch.no_grad():
        bev, feats = first.extract_feat(ex)
        for name in ("conv1", "conv2", "conv3"):       # the index side is dtype independent: exact
            assert int(feats[name].num_active_host()) == int(g[f"{name}_count"])
        for k in range(len(bev)):
            r = rel(cells_of(bev[k], g[f"bev{k}_cells"]), g[f"bev{k}_vals"])
            assert r <= (3e-2 if len(bev) == 1 else 4e-2), (k, r)
        preds = first.bbox_head(bev)
        for t, pd in enumerate(preds):
            for name in [k for k in pd.keys() if f"pred{t}_{k}_cells" in g.files]:
                v = pd[name]
                r = rel(cells_of(v, g[f"pred{t}_{name}_cells"]), g[f"pred{t}_{name}_vals"])
                assert r <= 6e-2, (t, name, r)
        out = first(ex, return_loss=False)
    # detections: a random-init head's scores sit within 1e-3 of each other at this size, so which cells
    # clear the pre_max cut under bf16 noise says nothing (the reduced-range bf16 tests compare detections);
    # here the output contract: caps, threshold, labels, every box finite and inside the limit range
    d = out[0]
    n = d["box3d_lidar"].shape[0]
    assert 0 < n <= g["det_boxes"].shape[0] + 8 and d["box3d_lidar"].shape[1] == g["det_boxes"].shape[1]
    assert (d["scores"] > 0.1).all() and torch.isfinite(d["box3d_lidar"]).all()
    assert int(d["label_preds"].min()) >= 0 and int(d["label_preds"].max()) <= int(g["det_labels"].max())
    lim = first.test_cfg["post_center_limit_range"]
    ctr = d["box3d_lidar"][:, :3].cpu().numpy()
    assert (ctr >= np.asarray(lim[:3], np.float32)).all() and (ctr <= np.asarray(lim[3:], np.float32)).all()
    first.to_inference(torch.float32)


# Bars, against what scripts/bf16_match_rate.py measured on MI355X (nusc / nusc1 / waymo / rcnn):
#   fp32, partner = same label, centre <= 5 cm, score <= 2e-2:                  1.000 on every case
#   bf16, same label + centre <= 5 cm (the box is there):                        0.978 / 0.976 / 0.966 / 0.975
#   bf16, ... and score <= 5e-2:                                                 0.956 / 0.966 / 0.916 / 0.964
#   bf16, ... and score <= 2e-2 (printed, not asserted):                         0.795 / 0.843 / 0.714 / 0.783
# The boxes without a partner sit at the post-NMS cap (the 83rd / 200th score of a task); centres of the others agree to
# <= 1.5 cm.  The bf16 score error (median 0.007, 90 % <= 0.03-0.036, max 0.06-0.09) is the 1 % relative error of the bf16
# BEV map seen through heat-map logits that span about +-10.
FP32_BAR = 0.98
BF16_PRESENT_BAR = 0.95
BF16_SCORE_TOL, BF16_BAR = 5e-2, 0.90


@pytest.mark.parametrize("kind", KINDS)
def test_calibrated_model_detections_match_oracle_in_fp32_and_bf16(cuda, kind):
    """Detection-level parity at full size on a model that behaves numerically like a trained one (BatchNorm statistics
    calibrated on the frame, head outputs of O(1), peaked heat maps: fullsize_cases.apply_calibration) against the fp32
    oracle's detections on that model (tests/golden/fullsize_<kind>_calibrated.npz).  fp32: every oracle detection has
    a partner (same label, centre within 5 cm, score within 2e-2; bar 98 %).  bf16 -- the benchmarked dtype, through
    PillarNet.forward (ranked lazy head) and, for the benchmarked nuScenes frame, through the two-lane frame graphs
    (bit-identical to eager) -- >= 95 % of the oracle's boxes are present (label, centre within 5 cm) and >= 90 % also
    agree in score to 5e-2; all rates are printed.  Also bounds the bf16 BEV map error on this model (3e-2)."""
    from pillarnet_lts_amd import FrameGraphs
    _cache.clear()
    cfg, model, first_cfg, first, pts = build_model(kind)
    g = np.load(os.path.join(G, f"fullsize_{kind}_calibrated.npz"))
    assert np.array_equal(sha(pts), g["points_sha"]) and np.array_equal(state_sha(model.state_dict()), g["weights_sha"])
    apply_calibration(first, g)
    want = dict(box3d_lidar=g["det_boxes"], scores=g["det_scores"], label_preds=g["det_labels"])
    assert want["scores"].shape[0] >= 200 and float(np.quantile(want["scores"], 0.9)) > 0.5      # a spread-out score field
    ex = {"points": [torch.from_numpy(pts).to(cuda)], "metadata": [None]}
    host = lambda d: {k: d[k].float().cpu().numpy() if k != "label_preds" else d[k].cpu().numpy()
                      for k in ("box3d_lidar", "scores", "label_preds")}
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-6))
    with torch.no_grad():
        first.to_inference(torch.float32)
        r32, hit, n = partner_rate(host(first(ex, return_loss=False)[0]), want)
        print(f"\n[{kind}] fp32 product vs fp32 oracle: {hit}/{n} = {r32:.3f}")
        assert r32 >= FP32_BAR, (kind, hit, n)
        first.to_inference(torch.bfloat16)
        bev, _ = first.extract_feat(ex)
        for k in range(len(bev)):
            r = rel(cells_of(bev[k], g[f"bev{k}_cells"]), g[f"bev{k}_vals"])
            print(f"[{kind}] bf16 BEV map {k}: relative L2 {r:.4f}")
            assert r <= 3e-2, (kind, k, r)
        eager = first(ex, return_loss=False)[0]
        r16, hit, n = partner_rate(host(eager), want, score_tol=BF16_SCORE_TOL)
        rp, hitp, _ = partner_rate(host(eager), want, score_tol=2.0)
        print(f"[{kind}] bf16 product vs fp32 oracle: present {hitp}/{n} = {rp:.3f}; score within {BF16_SCORE_TOL}: {hit}/{n} = {r16:.3f}")
        assert rp >= BF16_PRESENT_BAR, (kind, hitp, n)
        r16s, hits, _ = partner_rate(host(eager), want, score_tol=2e-2)
        print(f"[{kind}] bf16 product vs fp32 oracle, score within 2e-2: {hits}/{n} = {r16s:.3f}")
        assert r16 >= BF16_BAR, (kind, hit, n)
        if kind == "nusc":               # the path bench.py times: HIP-graph replays, two frames in flight
            graphs = FrameGraphs(first, lanes=2)
            handles = [graphs.forward_deferred(ex) for _ in range(4)]
            for h in handles:
                out = h.result()[0]
                for key in ("box3d_lidar", "scores", "label_preds"):
                    assert torch.equal(out[key], eager[key]), key
            print(f"[{kind}] bf16 two-lane frame graphs: bit-identical to eager, rate {r16:.3f}")
    del first, model


def test_pillarrcnn_second_stage_full_size_matches_oracle(cuda):
    """configs[4] at full size, fp32: the oracle's second stage (map fusion, RoI grid pooling, RoI MLPs,
    post_process; host) on the product's own first-stage maps and proposals against the product's output."""
    from oracle import rcnn_oracle as ro
    cfg, model, first_cfg, first, pts, g = case("rcnn")
    model.to_inference(torch.float32)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    ex = {"points": [torch.from_numpy(pts).to(cuda)], "metadata": [{"token": 0}]}
    seen = {}
    hook = model.second_stage[0].register_forward_pre_hook(lambda m, args: seen.update(args[0]))
    with torch.no_grad():
        out = model(ex, return_loss=False)
    hook.remove()
    n = int((seen["roi_labels"][0] > 0).sum())
    assert 100 < n <= 500 and seen["rois"].shape == (1, 500, 7)
    rois = seen["rois"].cpu().numpy()
    bev_map = np.ascontiguousarray(seen["bev_feature"].float().cpu().numpy())
    conv3 = np.ascontiguousarray(seen["backbone_features"]["conv3"].dense().float().cpu().numpy())
    fmap = ro.bev_stride_feature(sd, bev_map, [conv3])
    rng = first_cfg["reader"]["pc_range"]
    pooled, _ = ro.roi_grid_pool(rois, fmap, 7, 4 * cfg.pillar_size, rng[0], rng[1])
    cls, boxes = ro.roi_head(sd, pooled.reshape(1, 500, -1), rois)
    want = ro.post_process(cls, boxes, seen["roi_scores"].cpu().numpy(), seen["roi_labels"].cpu().numpy())[0]
    got = out[0]
    assert got["box3d_lidar"].shape == (n, 7)
    np.testing.assert_array_equal(got["label_preds"].cpu().numpy(), want["label_preds"])
    np.testing.assert_allclose(got["scores"].cpu().numpy(), want["scores"], atol=2e-4)
    np.testing.assert_allclose(got["box3d_lidar"].cpu().numpy(), want["box3d_lidar"], atol=8e-4, rtol=1e-3)
    assert np.abs(got["box3d_lidar"].cpu().numpy() - rois[0, :n]).max() > 1e-2
    _cache.clear()
