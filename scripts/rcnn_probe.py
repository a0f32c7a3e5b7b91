"""Probe: Waymo two-stage Pillar R-CNN (configs[4]) and one-stage PillarNet-34 FPN (configs[2]) frame times."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pillarnet_lts_amd import presets, synth

dev = torch.device("cuda:0")
torch.manual_seed(0)
pts = torch.from_numpy(synth.waymo_frame(180000, seed=100)).to(dev)
ex = {"points": [pts], "metadata": [None]}
ev = lambda: torch.cuda.Event(enable_timing=True)
for name in sys.argv[1:] or ["waymo_pillarrcnn_fpn", "waymo_pillarnet34_fpn"]:
    model = presets.build(getattr(presets, name)())
    synth.randomize_bn_(model, seed=1)
    model.to_inference(torch.bfloat16)
    with torch.no_grad():
        for _ in range(3):
            out = model(ex, return_loss=False)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10):
            out = model(ex, return_loss=False)
        torch.cuda.synchronize(); ms = (time.time() - t0) * 100
        print(name, "ms/frame", round(ms, 3), "boxes", tuple(out[0]["box3d_lidar"].shape), flush=True)
        if name == "waymo_pillarrcnn_fpn":
            e = [ev() for _ in range(6)]
            e[0].record()
            first, bev, feats, _ = model.single_det.forward_two_stage(ex, False)
            e[1].record()
            d = model.reorder_first_stage_prediction(first, {"batch_size": 1, "metadata": [None]})
            d["bev_feature"], d["backbone_features"] = bev[-1], feats
            e[2].record()
            fm = model.second_stage[0].fused_map(d)
            e[3].record()
            d = model.second_stage[0](d)
            e[4].record()
            d = model.point_head(d)
            o = model.post_process(model.roi_head(d, training=False))
            e[5].record()
            torch.cuda.synchronize()
            names = ["first_stage", "reorder", "fused_map", "bev_stride(all)", "heads+post"]
            print({n: round(e[i].elapsed_time(e[i + 1]), 3) for i, n in enumerate(names)}, flush=True)
