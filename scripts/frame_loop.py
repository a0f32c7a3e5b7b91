"""Runs the bf16 nuScenes PillarNet-18 frame a few times (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pillarnet_lts_amd import presets, synth

dev = torch.device("cuda:0")
torch.manual_seed(0)
dtype = torch.bfloat16 if len(sys.argv) < 2 or sys.argv[1] != "fp32" else torch.float32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
model = presets.build(presets.nusc_pillarnet18())
synth.randomize_bn_(model, seed=1)
model.to_inference(dtype)
pts = torch.from_numpy(synth.nusc_frame(10, 30000, seed=0)).to(dev)
ex = {"points": [pts], "metadata": [{}]}
with torch.no_grad():
    for it in range(iters):
        out = model(ex, return_loss=False)
torch.cuda.synchronize()
print("done", out[0]["box3d_lidar"].shape)
