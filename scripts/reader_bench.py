"""Device time of the reader (pn_pillarize: fill, mark, occupancy words + emit, PFN + scatter-max) on the nuScenes
10-sweep frame, replayed from a HIP graph; PN_PFN_VARIANT etc. in the environment select tuning variants."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pillarnet_lts_amd import presets, synth  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = presets.build(presets.nusc_pillarnet18())
synth.randomize_bn_(model, 1)
model.to_inference(torch.bfloat16)
ex = {"points": [torch.from_numpy(synth.nusc_frame(10, 30000, seed=0)).to(dev)], "metadata": [None]}
with torch.no_grad():
    sp = model.reader(ex)
    torch.cuda.synchronize()
    ref = sp.features.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sp = model.reader(ex)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    same = torch.equal(sp.features, ref)
print(f"variant {os.environ.get('PN_PFN_VARIANT', '0')}: reader {e0.elapsed_time(e1) * 1e3 / 50:.2f} us per call, "
      f"{sp.num_active_host()} pillars, features bit-identical to the first call: {same}", flush=True)
