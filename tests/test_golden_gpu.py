"""GPU: the PRODUCT's dense neck / CenterHead / predict against outputs the reference's own modules
produced (tests/golden/*.npz, see make_golden.py) -- same weights, same inputs."""
import json
import os

import numpy as np
import pytest
import torch

import logging
from pillarnet_lts_amd import necks
from pillarnet_lts_amd.bbox_heads import CenterHead
from pillarnet_lts_amd.center_utils import set_by_task_cfg
from pillarnet_lts_amd.config import ConfigDict

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
TASKS = [dict(stride=8, class_names=["car"]), dict(stride=8, class_names=["bus", "trailer"])]
HEADS = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)}


def _load_sd(module, g, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def test_rpn_necks_match_reference_outputs(cuda):
    g = np.load(os.path.join(G, "dense_neck.npz"))
    log = logging.getLogger("t")
    neck = necks.RPNV1(layer_nums=[2, 1], num_filters=24, in_channels=[16, 8], logger=log)
    _load_sd(neck, g, "rpnv1.sd.")
    neck.to(cuda).eval()
    t = lambda k: torch.from_numpy(g[k]).to(cuda)
    with torch.no_grad():
        y = neck({"conv4": t("rpnv1.in.conv4"), "conv5": t("rpnv1.in.conv5")})
    np.testing.assert_allclose(y[0].cpu().numpy(), g["rpnv1.out.0"], rtol=1e-4, atol=1e-4)
    ng = necks.RPNG(layer_nums=[1, 2], in_channels=[16, 8, 8], num_filters=[12, 8], logger=log)
    _load_sd(ng, g, "rpng.sd.")
    ng.to(cuda).eval()
    with torch.no_grad():
        y = ng({"conv3": t("rpng.in.conv3"), "conv4": t("rpng.in.conv4"), "conv5": t("rpng.in.conv5")})
    np.testing.assert_allclose(y[0].cpu().numpy(), g["rpng.out.0"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(y[1].cpu().numpy(), g["rpng.out.1"], rtol=1e-4, atol=1e-4)


def _head(cuda):
    g = np.load(os.path.join(G, "dense_head.npz"))
    head = CenterHead(tasks=[ConfigDict(t) for t in TASKS], in_channels=[24], code_weights=[1.0] * 10,
                      common_heads=HEADS, share_channel=16, reg_iou="GIoU", pillar_size=0.075,
                      point_cloud_range=[-6, -4.8, -5.0, 6, 4.8, 3.0])
    _load_sd(head, g, "sd.")
    return head.to(cuda).eval(), g


def test_center_head_forward_matches_reference(cuda):
    head, g = _head(cuda)
    with torch.no_grad():
        preds = head((torch.from_numpy(g["in.0"]).to(cuda),))
    for t, pd in enumerate(preds):
        for k, v in pd.items():
            np.testing.assert_allclose(v.cpu().numpy(), g[f"out.{t}.{k}"], rtol=1e-4, atol=1e-4)


def test_predict_matches_reference_predict(cuda):
    """The fused HIP decode + NMS on the reference's own head maps reproduces the reference's
    CenterHead.predict output: same boxes, labels, order."""
    head, g = _head(cuda)
    d = np.load(os.path.join(G, "decode.npz"))
    with open(os.path.join(G, "decode_test_cfg.json")) as f:
        test_cfg = ConfigDict(json.load(f))
    preds = [{k: torch.from_numpy(g[f"out.{t}.{k}"]).to(cuda) for k in list(HEADS) + ["hm"]} for t in range(2)]
    out = head.predict({"metadata": [None, None]}, preds, test_cfg)
    for b in range(2):
        np.testing.assert_array_equal(out[b]["label_preds"].cpu().numpy(), d[f"{b}.label_preds"])
        np.testing.assert_allclose(out[b]["scores"].cpu().numpy(), d[f"{b}.scores"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(out[b]["box3d_lidar"].cpu().numpy(), d[f"{b}.box3d_lidar"], rtol=1e-5, atol=1e-5)


def test_predict_double_flip_matches_reference(cuda):
    """test_cfg.double_flip: the four views of every sample are merged (maps flipped back, signs fixed, averaged
    after sigmoid / exp / IoU clamp) before the fused decode -- against the reference's own predict on an IoU
    head (its double-flip branch raises for heads without an IoU output)."""
    d = np.load(os.path.join(G, "decode.npz"))
    with open(os.path.join(G, "decode_df_test_cfg.json")) as f:
        test_cfg = ConfigDict(json.load(f))
    heads = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "iou": (1, 2)}
    head = CenterHead(tasks=[dict(t) for t in TASKS], in_channels=[24], code_weights=[1.0] * 8, common_heads=heads,
                      share_channel=16, reg_iou="GIoU", pillar_size=0.075,
                      point_cloud_range=[-6, -4.8, -5.0, 6, 4.8, 3.0]).to(cuda).eval()
    preds = [{k: torch.from_numpy(d[f"df.in.{t}.{k}"]).to(cuda) for k in list(heads) + ["hm"]} for t in range(2)]
    out = head.predict({"metadata": [{"token": i} for i in range(8)]}, preds, test_cfg)
    assert len(out) == 2
    for b in range(2):
        assert out[b]["metadata"]["token"] == int(d[f"df.{b}.token"]) == 4 * b
        np.testing.assert_array_equal(out[b]["label_preds"].cpu().numpy(), d[f"df.{b}.label_preds"])
        np.testing.assert_allclose(out[b]["scores"].cpu().numpy(), d[f"df.{b}.scores"], rtol=0, atol=5e-6)
        np.testing.assert_allclose(out[b]["box3d_lidar"].cpu().numpy(), d[f"df.{b}.box3d_lidar"], rtol=2e-5, atol=2e-5)
        assert out[b]["box3d_lidar"].shape[1] == 7


def test_predict_circular_nms_matches_reference(cuda):
    """test_cfg.circular_nms: centre-distance suppression (the reference's numba `circle_nms` on the CPU) through the
    same mask + greedy-select kernels, selected by a negative threshold."""
    head, g = _head(cuda)
    d = np.load(os.path.join(G, "decode.npz"))
    with open(os.path.join(G, "decode_circle_test_cfg.json")) as f:
        test_cfg = ConfigDict(json.load(f))
    preds = [{k: torch.from_numpy(g[f"out.{t}.{k}"]).to(cuda) for k in list(HEADS) + ["hm"]} for t in range(2)]
    out = head.predict({"metadata": [None, None]}, preds, test_cfg)
    for b in range(2):
        np.testing.assert_array_equal(out[b]["label_preds"].cpu().numpy(), d[f"circle.{b}.label_preds"])
        np.testing.assert_allclose(out[b]["scores"].cpu().numpy(), d[f"circle.{b}.scores"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(out[b]["box3d_lidar"].cpu().numpy(), d[f"circle.{b}.box3d_lidar"], rtol=1e-5, atol=1e-5)
    assert out[0]["scores"].shape[0] == 35          # the per-task caps 20 + 15 are both reached
