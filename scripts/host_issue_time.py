"""Host time to queue one frame's kernels vs time spent waiting for the GPU, in the pipelined loop.
If the wait is ~0 the loop is host-bound (the GPU drains the queue faster than Python fills it)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pillarnet_lts_amd import presets, synth

torch.manual_seed(0)
cfg = presets.nusc_pillarnet18()
model = presets.build(cfg)
synth.randomize_bn_(model, seed=1)
model.to_inference(torch.bfloat16)
frames = [torch.from_numpy(synth.nusc_frame(10, 30000, seed=i)).cuda() for i in range(4)]
ex = [{"points": [f], "metadata": [{"token": i}]} for i, f in enumerate(frames)]
with torch.no_grad():
    for i in range(8):
        model(ex[i % 4], return_loss=False)
    torch.cuda.synchronize()
    n = 60
    issue = wait = 0.0
    t_all = time.perf_counter()
    pending = None
    for i in range(n):
        t0 = time.perf_counter()
        nxt = model.forward_deferred(ex[i % 4])
        t1 = time.perf_counter()
        if pending is not None:
            pending.result()
        t2 = time.perf_counter()
        issue += t1 - t0
        wait += t2 - t1
        pending = nxt
    pending.result()
    torch.cuda.synchronize()
    total = time.perf_counter() - t_all
    print(f"per frame: total {1e6 * total / n:.1f} us  host issue {1e6 * issue / n:.1f} us  result() {1e6 * wait / n:.1f} us")
    # host issue time alone: queue frames without ever waiting (GPU falls behind, host never blocks)
    t0 = time.perf_counter()
    hs = [model.forward_deferred(ex[i % 4]) for i in range(20)]
    t1 = time.perf_counter()
    for h in hs:
        h.result()
    print(f"host issue only (20 frames queued back to back): {1e6 * (t1 - t0) / 20:.1f} us/frame")

    # the same through frame graphs: host time of a replay (points copy + graph launch + output clones + count copy)
    from pillarnet_lts_amd import FrameGraphs
    for lanes in (1, 2):
        g = FrameGraphs(model, lanes=lanes)
        for i in range(2 * lanes):
            g.forward_deferred(ex[i % 4]).result()
        torch.cuda.synchronize()
        issue = wait = 0.0
        inflight = []
        n = 200
        t_all = time.perf_counter()
        for i in range(n):
            t0 = time.perf_counter()
            inflight.append(g.forward_deferred(ex[i % 4]))
            t1 = time.perf_counter()
            if len(inflight) > lanes:
                inflight.pop(0).result()
            t2 = time.perf_counter()
            issue += t1 - t0
            wait += t2 - t1
        for h in inflight:
            h.result()
        total = time.perf_counter() - t_all
        print(f"graphs, {lanes} lane(s): per frame total {1e6 * total / n:.1f} us  host issue {1e6 * issue / n:.1f} us  result() {1e6 * wait / n:.1f} us")
        t0 = time.perf_counter()
        hs = [g.forward_deferred(ex[i % 4]) for i in range(20)]
        t1 = time.perf_counter()
        for h in hs:
            h.result()
        print(f"graphs, {lanes} lane(s): host issue only (20 frames back to back) {1e6 * (t1 - t0) / 20:.1f} us/frame")
        fr = next(iter(g.lanes[0].frames.values()))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fr.graph.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"graphs, {lanes} lane(s): graph.replay() alone, host side {1e6 * (t1 - t0) / 20:.1f} us")
