"""One frame at a time through a one-lane frame graph: device time of the bare replay (back to back, no host in between)
against the loop that hands every frame's detections to the host, and the host-side pieces of that hand-over."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pillarnet_lts_amd import FrameGraphs, presets, synth

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = presets.build(presets.nusc_pillarnet18())
synth.randomize_bn_(model, 1)
model.to_inference(torch.bfloat16)
exs = [{"points": [torch.from_numpy(synth.nusc_frame(10, 30000, seed=i)).to(dev)], "metadata": [{"token": i}]} for i in range(4)]
with torch.no_grad():
    # ONE_LANE_RULES_AHEAD=0 / 1: the backbone's index chain in line / on the graph's side branch (default: FrameGraphs' own)
    ra = os.environ.get("ONE_LANE_RULES_AHEAD")
    g = FrameGraphs(model, lanes=1, rules_ahead=None if ra is None else bool(int(ra)))
    for e in exs:
        g(e)
    fr = next(iter(g.lanes[0].frames.values()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fr.graph.replay()
    e1.record()
    torch.cuda.synchronize()
    bare = e0.elapsed_time(e1) / 100
    t0 = time.perf_counter()
    for i in range(200):
        out = g(exs[i % 4])
    torch.cuda.synchronize()
    loop = (time.perf_counter() - t0) / 200 * 1e3
    # host pieces
    t0 = time.perf_counter()
    for i in range(200):
        h = g.forward_deferred(exs[i % 4])
    t_queue = (time.perf_counter() - t0) / 200 * 1e3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        h.out = None
        h.result()
    t_result = (time.perf_counter() - t0) / 200 * 1e3
print(f"bare replay {bare * 1e3:.1f} us/frame   loop with the host taking every frame {loop * 1e3:.1f} us/frame   "
      f"host: queue a frame {t_queue * 1e3:.1f} us, result() on a finished frame {t_result * 1e3:.1f} us", flush=True)
