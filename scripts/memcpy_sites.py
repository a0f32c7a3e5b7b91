"""Where do device-to-device copies come from?  torch.profiler with stacks over a few frames of a preset."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from pillarnet_lts_amd import presets, synth

name = sys.argv[1] if len(sys.argv) > 1 else "waymo_pillarnet34_fpn"
dev = torch.device("cuda:0")
torch.manual_seed(0)
pts = torch.from_numpy(synth.waymo_frame(180000, seed=100) if "waymo" in name else synth.nusc_frame(10, 30000, seed=0)).to(dev)
ex = {"points": [pts], "metadata": [None]}
model = presets.build(getattr(presets, name)())
synth.randomize_bn_(model, seed=1)
model.to_inference(torch.bfloat16)
with torch.no_grad():
    for _ in range(3):
        model(ex, return_loss=False)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(2):
            model(ex, return_loss=False)
        torch.cuda.synchronize()
ops = {}
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::index", "aten::cat", "aten::zeros", "aten::fill_", "aten::zero_"):
        st = [s for s in (e.stack or []) if "pillarnet" in s]
        key = (e.name, st[0] if st else "?")
        ops[key] = ops.get(key, 0) + 1
for k, v in sorted(ops.items(), key=lambda kv: -kv[1])[:40]:
    print(v, k)
