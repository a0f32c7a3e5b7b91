"""Probe: full detector forward on the GPU, fp32 and bf16, with per-stage timings."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pillarnet_lts_amd as P
from pillarnet_lts_amd import presets, synth

dev = torch.device("cuda:0")
torch.manual_seed(0)
for dtype in (torch.float32, torch.bfloat16):
    cfg = presets.nusc_pillarnet18()
    model = presets.build(cfg)
    synth.randomize_bn_(model, seed=1)
    model.to_inference(dtype)
    pts = torch.from_numpy(synth.nusc_frame(10, 30000, seed=0)).to(dev)
    ex = {"points": [pts], "metadata": [{}]}
    with torch.no_grad():
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            out = model(ex, return_loss=False)
            torch.cuda.synchronize(); t1 = time.time()
            print(dtype, "iter", it, "ms", (t1 - t0) * 1e3, "boxes", out[0]["box3d_lidar"].shape,
                  "labels", out[0]["label_preds"][:5].tolist(), flush=True)
        # stage timings
        ev = lambda: torch.cuda.Event(enable_timing=True)
        e = [ev() for _ in range(6)]
        e[0].record()
        sp = model.reader({"points": [pts]})
        if dtype != torch.float32:
            sp = sp.replace_feature(sp.features_cap.to(dtype))
        e[1].record()
        feats = model.backbone(sp)
        e[2].record()
        bev = model.neck(feats)
        e[3].record()
        preds = model.bbox_head(bev)
        e[4].record()
        det = model.bbox_head.predict(ex, preds, model.test_cfg)
        e[5].record()
        torch.cuda.synchronize()
        names = ["reader", "backbone", "neck", "head", "predict"]
        print({n: round(e[i].elapsed_time(e[i + 1]), 3) for i, n in enumerate(names)})
        print("M per level:", sp.num_active_host(), [feats[k].num_active_host() for k in ("conv1", "conv2", "conv3")])
        print("scores", det[0]["scores"][:8].tolist())
