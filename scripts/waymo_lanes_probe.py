"""Waymo PillarNet-34 FPN (configs[2]) and two-stage Pillar R-CNN (configs[4]) frame rates through FrameGraphs with two
lanes, beside the eager loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pillarnet_lts_amd import FrameGraphs, presets, synth

dev = torch.device("cuda:0")
torch.manual_seed(0)
frames = [torch.from_numpy(synth.waymo_frame(180000, seed=100 + i)).to(dev) for i in range(4)]
exs = [{"points": [p], "metadata": [None]} for p in frames]
for name in sys.argv[1:] or ["waymo_pillarnet34_fpn", "waymo_pillarrcnn_fpn"]:
    model = presets.build(getattr(presets, name)())
    synth.randomize_bn_(model, seed=1)
    model.to_inference(torch.bfloat16)
    with torch.no_grad():
        for e in exs:
            model(e, return_loss=False)
        torch.cuda.synchronize(); t0 = time.time()
        for i in range(40):
            model(exs[i % 4], return_loss=False)
        torch.cuda.synchronize(); eager = (time.time() - t0) / 40 * 1e3
        g = FrameGraphs(model, lanes=2)
        for e in exs:
            g.forward_deferred(e).result(); g.forward_deferred(e).result()
        torch.cuda.synchronize(); t0 = time.time()
        inflight = []
        for i in range(200):
            inflight.append(g.forward_deferred(exs[i % 4]))
            if len(inflight) > 2:
                out = inflight.pop(0).result()
        for h in inflight:
            out = h.result()
        torch.cuda.synchronize(); lanes = (time.time() - t0) / 200 * 1e3
    print(f"{name}: eager {eager:.3f} ms/frame, two lanes {lanes:.3f} ms/frame ({1e3 / lanes:.0f} frames/s), boxes {tuple(out[0]['box3d_lidar'].shape)}", flush=True)
    del g, model
