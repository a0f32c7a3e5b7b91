"""CPU: the restated training pieces (losses, CenterHead.loss, gaussian targets, one-cycle schedule)
against outputs of the reference's own code (tests/golden/train.npz, make_golden.py::golden_train),
plus a world-size-2 gloo run of the gradient all-reduce."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from pillarnet_lts_amd import losses, solver, targets
from pillarnet_lts_amd.bbox_heads import CenterHead
from pillarnet_lts_amd.config import ConfigDict

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "train.npz"))
TASKS = [dict(stride=8, class_names=["car"]), dict(stride=8, class_names=["bus", "trailer"])]
HEADS = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)}
RANGE = [-6, -4.8, -5.0, 6, 4.8, 3.0]


def test_gaussian_helpers():
    for a, r1, r5 in zip(G["gr.sizes"], G["gr.r01"], G["gr.r05"]):
        assert abs(targets.gaussian_radius(tuple(a), 0.1) - r1) < 1e-9
        assert abs(targets.gaussian_radius(tuple(a), 0.5) - r5) < 1e-9
    hm = np.zeros((40, 50), np.float32)
    for (x, y, r) in [(5.3, 7.9, 2), (48.2, 38.5, 4), (0.1, 0.2, 3), (25.0, 20.0, 6)]:
        targets.draw_gaussian(hm, np.array([x, y], np.float32), r)
    np.testing.assert_array_equal(hm, G["gauss.hm"])


def test_assign_labels_matches_reference():
    out = targets.assign_labels(G["al.boxes"], G["al.classes"], TASKS, RANGE, 0.075, 0.1, 50, 2)
    for key in ("hm", "anno_box", "ind", "mask", "cat", "gt_box"):
        for t in range(2):
            want = G[f"al.{key}.{t}"]
            got = np.asarray(out[key][t])
            assert got.shape == want.shape, (key, t)
            if got.dtype.kind == "f":
                np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
            else:
                np.testing.assert_array_equal(got, want)
    assert sum(int(G[f"al.mask.{t}"].sum()) for t in range(2)) >= 8
    # the two-stage target (preprocess.py:324-343): all boxes task by task, heading at column 6, class last
    np.testing.assert_allclose(out["gt_boxes_and_cls"], G["al.gt_boxes_and_cls"], rtol=1e-6, atol=1e-6)


def _t(k):
    return torch.from_numpy(G[k])


def test_losses_match_reference():
    ind, mask, cat = _t("fl.ind"), _t("fl.mask"), _t("fl.cat")
    fl = losses.fast_focal_loss(_t("fl.out"), _t("fl.target"), ind, mask, cat)
    np.testing.assert_allclose(fl.numpy(), G["fl.loss"], rtol=1e-6)
    rl = losses.reg_loss(_t("rl.out"), mask, ind, _t("rl.target"))
    np.testing.assert_allclose(rl.numpy(), G["rl.loss"], rtol=1e-5, atol=1e-7)
    for kind in ("IoU", "GIoU", "DIoU"):
        il = losses.iou_reg_loss(_t("il.pred"), mask, ind, _t("il.gt"), kind)
        np.testing.assert_allclose(il.numpy(), G[f"il.{kind}"], rtol=1e-5)
    z = torch.zeros_like(mask)
    assert float(losses.reg_loss(_t("rl.out"), z, ind, _t("rl.target")).abs().sum()) == 0.0
    assert float(losses.iou_reg_loss(_t("il.pred"), z, ind, _t("il.gt"), "GIoU")) == 0.0
    fl0 = losses.fast_focal_loss(_t("fl.out"), _t("fl.target"), ind, z, cat)      # no positives: the negative term alone
    neg = (torch.log(1 - _t("fl.out")) * _t("fl.out").pow(2) * (1 - _t("fl.target")).pow(4)).sum()
    np.testing.assert_allclose(fl0.numpy(), (-neg).numpy(), rtol=1e-6)


def test_center_head_loss_matches_reference():
    h = np.load(os.path.join(os.path.dirname(__file__), "golden", "dense_head.npz"))
    head = CenterHead(tasks=[ConfigDict(t) for t in TASKS], in_channels=[24], code_weights=[1.0] * 10,
                      common_heads=HEADS, share_channel=16, reg_iou="GIoU", pillar_size=0.075, point_cloud_range=RANGE)
    head.load_state_dict({k[3:]: torch.from_numpy(h[k]) for k in h.files if k.startswith("sd.")})
    head.eval()
    with torch.no_grad():
        preds = head((_t("hl.x"),))
    ex = {key: [torch.from_numpy(np.stack([G[f"al.{key}.{t}"]] * 2)) for t in range(2)]
          for key in ("hm", "anno_box", "ind", "mask", "cat", "gt_box")}
    out = head.loss(ex, preds, dict(hm_weight=1, bbox_weight=0.25, iou_weight=1, reg_iou_weight=0.25))
    for key in ("loss", "hm_loss", "loc_loss", "loc_loss_elem", "num_positive", "reg_iou_loss"):
        for t in range(2):
            np.testing.assert_allclose(np.asarray(out[key][t].detach()), G[f"hl.{key}.{t}"], rtol=2e-5, atol=1e-6)


def test_one_cycle_matches_reference():
    sch = solver.OneCycle(200, 0.001, (0.95, 0.85), 10.0, 0.4)
    got = np.asarray([sch.at(s) for s in range(200)])
    np.testing.assert_allclose(got, G["oc.seq"], rtol=1e-12, atol=1e-15)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo")
    torch.manual_seed(0)
    lin = torch.nn.Linear(4, 3)
    x = torch.full((2, 4), float(rank + 1))
    lin(x).sum().backward()
    solver.allreduce_grads(lin.parameters())
    q.put((rank, lin.weight.grad.clone().numpy(), lin.bias.grad.clone().numpy()))
    torch.distributed.destroy_process_group()


def test_allreduce_grads_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=120) for _ in ps), key=lambda r: r[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    # rank r's weight grad is 2*(r+1) everywhere -> mean 3; bias grad 2 on both -> 2
    for _, gw, gb in res:
        np.testing.assert_allclose(gw, 3.0)
        np.testing.assert_allclose(gb, 2.0)


def _rcnn_train_golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "rcnn_train.npz"))


def _oracle_iou3d(a, b):
    """boxes_iou3d_gpu (iou3d_nms_utils.py:37-74) around the compiled-reference-pinned BEV overlap."""
    from oracle import iou3d
    from pillarnet_lts_amd.iou3d_nms import to_pcdet
    a, b = to_pcdet(a), to_pcdet(b)
    ov = torch.from_numpy(iou3d.iou_matrix(a.numpy(), b.numpy(), overlap_only=True)).float()
    hmax = torch.min((a[:, 2] + a[:, 5] / 2).view(-1, 1), (b[:, 2] + b[:, 5] / 2).view(1, -1))
    hmin = torch.max((a[:, 2] - a[:, 5] / 2).view(-1, 1), (b[:, 2] - b[:, 5] / 2).view(1, -1))
    o3 = ov * torch.clamp(hmax - hmin, min=0)
    va, vb = (a[:, 3] * a[:, 4] * a[:, 5]).view(-1, 1), (b[:, 3] * b[:, 4] * b[:, 5]).view(1, -1)
    return o3 / torch.clamp(va + vb - o3, min=1e-6)


RCNN_TRAIN_CFG = dict(CLASS_AGNOSTIC=True, SHARED_FC=[32, 32], CLS_FC=[32, 32], REG_FC=[32, 32], DP_RATIO=0.3,
                      TARGET_CONFIG=dict(ROI_PER_IMAGE=32, FG_RATIO=0.5, SAMPLE_ROI_BY_EACH_CLASS=True,
                                         CLS_SCORE_TYPE="roi_iou", CLS_FG_THRESH=0.7, CLS_BG_THRESH=0.25,
                                         CLS_BG_THRESH_LO=0.1, HARD_BG_RATIO=0.8, REG_FG_THRESH=0.5),
                      LOSS_CONFIG=dict(CLS_LOSS="BinaryCrossEntropy", REG_LOSS="L1",
                                       LOSS_WEIGHTS=dict(rcnn_cls_weight=1.0, rcnn_reg_weight=1.0,
                                                         code_weights=[1.0] * 7)))


def test_proposal_targets_and_roi_losses_match_reference():
    """ProposalTargetLayer + RoI head target encoding + losses against the reference's own modules run on CPU
    under the same numpy / torch seeds (tests/golden/rcnn_train.npz)."""
    from pillarnet_lts_amd import second_stage
    from pillarnet_lts_amd.config import ConfigDict
    g = _rcnn_train_golden()
    head = second_stage.RoIMIXHead(in_channels=4, model_cfg=ConfigDict(RCNN_TRAIN_CFG), code_size=7, mixer_type="",
                                   num_patches=49).train()
    head.proposal_target_layer.iou_fn = _oracle_iou3d
    t = lambda k: torch.from_numpy(g[k])
    batch = dict(batch_size=2, rois=t("in.rois"), roi_scores=t("in.roi_scores"), roi_labels=t("in.roi_labels"),
                 gt_boxes_and_cls=t("in.gt_boxes_and_cls"))
    np.random.seed(7)
    torch.manual_seed(8)
    td = head.assign_targets(batch)
    for k in ("rois", "roi_scores", "roi_labels", "reg_valid_mask", "gt_of_rois_src"):
        np.testing.assert_array_equal(td[k].numpy(), g["pt." + k], err_msg=k)
    for k in ("gt_iou_of_rois", "rcnn_cls_labels", "gt_of_rois"):
        np.testing.assert_allclose(td[k].numpy(), g["pt." + k], rtol=1e-6, atol=1e-6, err_msg=k)
    td["rcnn_cls"], td["rcnn_reg"] = t("loss.in.rcnn_cls"), t("loss.in.rcnn_reg")
    head.forward_ret_dict = td
    loss, tb = head.get_loss()
    np.testing.assert_allclose(float(loss), float(g["loss.total"]), rtol=1e-6)
    np.testing.assert_allclose(float(tb["rcnn_loss_cls"]), float(g["loss.cls"]), rtol=1e-6)
    np.testing.assert_allclose(float(tb["rcnn_loss_reg"]), float(g["loss.reg"]), rtol=1e-6)


def test_point_head_focal_loss_matches_reference():
    from pillarnet_lts_amd import second_stage
    from pillarnet_lts_amd.config import ConfigDict
    g = _rcnn_train_golden()
    ph = second_stage.PointHead(in_channels=4, num_class=1, model_cfg=ConfigDict(
        CLS_FC=[32, 32], LOSS_CONFIG=dict(LOSS_REG="smooth-l1", LOSS_WEIGHTS=dict(point_cls_weight=1.0))))
    ph.forward_ret_dict = dict(point_cls_preds=torch.from_numpy(g["ph.in.preds"]),
                               point_cls_labels=torch.from_numpy(g["ph.point_cls_labels"]))
    loss, tb = ph.get_loss()
    np.testing.assert_allclose(float(loss), float(g["ph.loss"]), rtol=1e-6)
    assert tb["point_pos_num"] == float(g["ph.pos_num"])


def test_masked_iou_losses_have_the_compacted_gradient():
    """the `where`-masked form (no boolean indexing, no host wait) against the reference's compacted form"""
    torch.manual_seed(3)
    B, K = 2, 12
    mask = (torch.rand(B, K) < 0.4).to(torch.uint8)
    mask[1] = 0                                            # a sample with no objects; padding rows are all-zero boxes
    ind = torch.randint(0, 64, (B, K))
    gt = torch.rand(B, K, 7) + 0.5
    gt = gt * mask.unsqueeze(-1).float()
    for kind in ("IoU", "GIoU", "DIoU"):
        a = (torch.rand(B, 8, 8, 7) + 0.5).requires_grad_()
        got = losses.iou_reg_loss(a, mask, ind, gt, kind)
        got.backward()
        b = a.detach().clone().requires_grad_()
        m = mask.bool()
        want = (1.0 - losses.axis_aligned_iou3d(losses.gather_at(b, ind)[m], gt[m], kind)).sum() / (m.sum() + 1e-4)
        want.backward()
        np.testing.assert_allclose(got.detach().numpy(), want.detach().numpy(), rtol=1e-6)
        assert torch.isfinite(a.grad).all()
        np.testing.assert_allclose(a.grad.numpy(), b.grad.numpy(), rtol=1e-5, atol=1e-7)


def _extra_golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "rcnn_extra.npz"))


def _oracle_aligned_iou3d(a, b):
    return torch.diagonal(_oracle_iou3d(a, b)).reshape(-1, 1).contiguous() if a.shape[0] else a.new_zeros((0, 1))


EXTRA_CFG = dict(CLASS_AGNOSTIC=True, TARGET_CONFIG=RCNN_TRAIN_CFG["TARGET_CONFIG"],
                 LOSS_CONFIG=dict(CLS_LOSS="BinaryCrossEntropy", REG_LOSS="L1", IOU_LOSS="L1",
                                  LOSS_WEIGHTS=dict(rcnn_cls_weight=1.0, rcnn_reg_weight=1.0, rcnn_iou_weight=0.7,
                                                    code_weights=[1.0] * 7)))


def _load_prefixed(module, g, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
    module.load_state_dict(sd, strict=True)      # strict: parameter names and shapes are the reference's
    return module


def test_roi_ffn_head_matches_reference():
    """ROI_HEAD.RoIFFNHead against the reference's own module (roi_ffn_head.py:10-140) run on CPU: inference outputs,
    then a training pass -- targets under the same seeds, forward, class / box / IoU-branch losses
    (roi_head_template.py:117-147) and two gradients.  mmcv is absent, so the fixture's FFN block was this package's
    restatement: that block alone is unpinned."""
    from pillarnet_lts_amd import second_stage
    from pillarnet_lts_amd.config import ConfigDict
    g, t = _extra_golden(), _rcnn_train_golden()
    head = _load_prefixed(second_stage.RoIFFNHead(model_cfg=ConfigDict(EXTRA_CFG), feedforward_channels=48, content_dim=32,
                                                  num_class=1, code_size=7), g, "sd.ffn.")
    assert head.use_iou
    tt = lambda a, k: torch.from_numpy(a[k]).clone()
    B = g["bevf.in.rois"].shape[0]
    head.eval()
    with torch.no_grad():
        out = head(dict(batch_size=B, rois=tt(g, "bevf.in.rois"), roi_features=tt(g, "ffn.in.roi_features"),
                        roi_scores=tt(g, "ffn.in.roi_scores"), roi_labels=tt(g, "ffn.in.roi_labels")), training=False)
    np.testing.assert_allclose(out["batch_cls_preds"].numpy(), g["ffn.out.batch_cls_preds"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["batch_box_preds"].numpy(), g["ffn.out.batch_box_preds"], rtol=1e-5, atol=1e-6)
    head.train()
    head.proposal_target_layer.iou_fn = _oracle_iou3d
    head.aligned_iou_fn = _oracle_aligned_iou3d
    np.random.seed(7)
    torch.manual_seed(8)
    td = head.assign_targets(dict(batch_size=2, rois=tt(t, "in.rois"), roi_scores=tt(t, "in.roi_scores"),
                                  roi_labels=tt(t, "in.roi_labels"), gt_boxes_and_cls=tt(t, "in.gt_boxes_and_cls")))
    np.testing.assert_array_equal(td["rois"].numpy(), t["pt.rois"])
    head(dict(batch_size=2, rois=td["rois"], roi_labels=td["roi_labels"], roi_scores=td["roi_scores"],
              roi_targets_dict=td, roi_features=tt(g, "ffn.train.in.roi_features")), training=True)
    fr = head.forward_ret_dict
    for k in ("rcnn_cls", "rcnn_reg", "rcnn_iou", "batch_box_preds"):
        np.testing.assert_allclose(fr[k].detach().numpy(), g["ffn.train." + k], rtol=1e-5, atol=1e-6, err_msg=k)
    loss, tb = head.get_loss()
    np.testing.assert_allclose(float(loss), float(g["ffn.train.loss"]), rtol=1e-5)
    for k in ("cls", "reg", "iou"):
        np.testing.assert_allclose(float(tb["rcnn_loss_" + k]), float(g["ffn.train.loss_" + k]), rtol=1e-5, err_msg=k)
    loss.backward()
    np.testing.assert_allclose(head.fc_iou.weight.grad.numpy(), g["ffn.train.grad.fc_iou.weight"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(head.ffn.layers[0][0].weight.grad.numpy(), g["ffn.train.grad.ffn.layers.0.0.weight"],
                               rtol=1e-4, atol=1e-6)


def test_legacy_roi_head_training_branch_matches_reference():
    """ROI_HEAD.RoIHead with training=True (roi_head.py:70-111 + the template's targets and losses) against the
    reference's own module on CPU."""
    from pillarnet_lts_amd import legacy
    from pillarnet_lts_amd.config import ConfigDict
    g, t = _extra_golden(), _rcnn_train_golden()
    cfg = ConfigDict(dict(EXTRA_CFG, SHARED_FC=[32, 32], CLS_FC=[32, 32], REG_FC=[32, 32], DP_RATIO=0.0))
    head = _load_prefixed(legacy.RoIHead(input_channels=32, model_cfg=cfg, num_class=1, code_size=7), g, "sd.roihead.").train()
    head.proposal_target_layer.iou_fn = _oracle_iou3d
    tt = lambda a, k: torch.from_numpy(a[k]).clone()
    np.random.seed(7)
    torch.manual_seed(8)
    td = head.assign_targets(dict(batch_size=2, rois=tt(t, "in.rois"), roi_scores=tt(t, "in.roi_scores"),
                                  roi_labels=tt(t, "in.roi_labels"), gt_boxes_and_cls=tt(t, "in.gt_boxes_and_cls")))
    head(dict(rois=td["rois"], roi_labels=td["roi_labels"], roi_scores=td["roi_scores"], roi_targets_dict=td,
              roi_features=tt(g, "ffn.train.in.roi_features")), training=True)
    for k in ("rcnn_cls", "rcnn_reg"):
        np.testing.assert_allclose(head.forward_ret_dict[k].detach().numpy(), g["roihead.train." + k], rtol=1e-4, atol=1e-5)
    loss, tb = head.get_loss()
    np.testing.assert_allclose(float(loss), float(g["roihead.train.loss"]), rtol=1e-5)
    np.testing.assert_allclose(float(tb["rcnn_loss_cls"]), float(g["roihead.train.loss_cls"]), rtol=1e-5)
    np.testing.assert_allclose(float(tb["rcnn_loss_reg"]), float(g["roihead.train.loss_reg"]), rtol=1e-5)


def test_bev_feature_builds_with_reference_names():
    """SECOND_STAGE.BEVFeature (bev_interpolation.py:17-159): constructor kwargs, lateral kinds and state-dict keys
    of the two layouts the fixture holds (forward outputs are checked on the GPU: the pooling kernel is HIP-only)."""
    from pillarnet_lts_amd import second_stage
    from pillarnet_lts_amd.registry import SECOND_STAGE, ROI_HEAD
    assert SECOND_STAGE.get("BEVFeature") is second_stage.BEVFeature and ROI_HEAD.get("RoIFFNHead") is second_stage.RoIFFNHead
    g = _extra_golden()
    for tag, out_stride in (("s4", 4), ("s8", 8)):
        bev = second_stage.BEVFeature(feature_sources=["conv4"], pillar_size=0.1, pc_range=[-6.4, -4.8, -2.0, 6.4, 4.8, 4.0],
                                      out_stride=out_stride, grid_size=5, in_channels=24, share_channels=16,
                                      backbone_channels=dict(conv1=32, conv2=64, conv3=128, conv4=256),
                                      backbone_strides=dict(conv1=1, conv2=2, conv3=4, conv4=8))
        _load_prefixed(bev, g, f"sd.bevf.{tag}.")
        assert bev.lat_tensor_type == ["dense"]
    # a source at the output stride (unless that stride is 8) takes the 1x1 sparse conv, a finer one a block conv
    bev = second_stage.BEVFeature(feature_sources=["conv1", "conv3", "conv4"], pillar_size=0.1, pc_range=[0, 0, -2, 4.8, 3.2, 2],
                                  out_stride=4, backbone_channels=dict(conv1=32, conv2=64, conv3=128, conv4=256),
                                  backbone_strides=dict(conv1=1, conv2=2, conv3=4, conv4=8))
    assert bev.lat_tensor_type == ["sparse", "sparse", "dense"]
    assert bev.lat_conv[0][0].weight.shape == (128, 4, 4, 32) and bev.lat_conv[1][0].weight.shape == (128, 1, 1, 128)
    assert bev.fusion_conv[0].in_channels == 4 * 128
