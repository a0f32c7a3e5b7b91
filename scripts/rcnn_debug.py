import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import rcnn_oracle as ro
from pillarnet_lts_amd import presets, synth
import test_rcnn_gpu as T
torch.manual_seed(3)
cfg = T.small_rcnn(); model = presets.build(cfg); synth.randomize_bn_(model, seed=4)
with torch.no_grad():
    for th in model.bbox_head.task_heads: th.hm[-1].weight.mul_(8.0)
    model.roi_head.reg_layers[-1].weight.mul_(100.0)
pts = [synth.waymo_frame(7000, seed=101)[:, :5] * np.array([0.17, 0.17, 1, 1, 1], np.float32)]
model.to_inference(torch.float32)
sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
ex = {"points": [torch.from_numpy(p).cuda() for p in pts], "metadata": [None]}
with torch.no_grad():
    first, bev, feats, _ = model.single_det.forward_two_stage(ex, False)
    e = model.reorder_first_stage_prediction(first, {"batch_size": 1, "metadata": [None]})
    e["bev_feature"], e["backbone_features"] = bev[-1], feats
    bsf = model.second_stage[0]
    fm = bsf.fused_map(e)
    e = bsf(e)
    e2 = model.roi_head(dict(e), training=False)
fmap = ro.bev_stride_feature(sd, np.ascontiguousarray(bev[-1].float().cpu().numpy()), [np.ascontiguousarray(feats["conv3"].dense().float().cpu().numpy())])
print("fused map max diff", np.abs(fm.cpu().numpy() - fmap).max(), "mag", np.abs(fmap).max())
rois = e["rois"].cpu().numpy()
pooled, _ = ro.roi_grid_pool(rois, fmap, 7, 0.4, -12.8, -12.8)
pooled_p, _ = ro.roi_grid_pool(rois, fm.cpu().numpy(), 7, 0.4, -12.8, -12.8)
gp = e["point_features"].cpu().numpy()
d = np.abs(gp - pooled).reshape(500, -1).max(1); d2 = np.abs(gp - pooled_p).reshape(500, -1).max(1)
print("pooled diff (oracle map) top", np.sort(d)[-5:], "argmax", np.argsort(d)[-5:])
print("pooled diff (product map) top", np.sort(d2)[-5:], np.argsort(d2)[-5:])
cls, boxes = ro.roi_head(sd, pooled.reshape(1, 500, -1), rois)
cls_p, boxes_p = ro.roi_head(sd, gp.reshape(1, 500, -1), rois)
g = e2["batch_cls_preds"].cpu().numpy()
dc = np.abs(g - cls).reshape(-1); print("cls diff top", np.sort(dc)[-5:], np.argsort(dc)[-5:])
dc = np.abs(g - cls_p).reshape(-1); print("cls diff (same pooled) top", np.sort(dc)[-5:], np.argsort(dc)[-5:])
i = int(np.argsort(d2)[-1]); print("roi", rois[0, i], "n", first[0]["box3d_lidar"].shape[0])
j = int(np.abs(gp - pooled_p).reshape(500, 49, -1)[i].max(1).argmax())
p = ro.grid_points(rois[0, i:i+1], 7)[0, j]; print("pt", p, (p + 12.8) / 0.4, (p+np.float32(12.8))*np.float32(1/0.4))
