"""Probe: time of one data-parallel training step (BASELINE configs[3] shapes, per-GPU batch 4, fp32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pillarnet_lts_amd import presets, solver, synth, targets

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = presets.nusc_pillarnet18()
model = presets.build(cfg).to(dev).train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = cfg.model.reader.pc_range
samples = [targets.synthetic_example(synth.nusc_frame(10, 30000, seed=s), cfg.model.bbox_head.tasks, rng,
                                     cfg.pillar_size, n_boxes=40, seed=s, train_cfg=cfg.train_cfg) for s in range(B)]
ex = targets.collate(samples, dev)
opt = solver.TrueWdAdam(model.parameters(), wd=0.01)
sch = solver.OneCycle(100, 1e-3)
for step in range(2):
    v, _ = solver.train_step(model, ex, opt, sch, step)
torch.cuda.synchronize(); t0 = time.time()
n = 3
for step in range(2, 2 + n):
    v, _ = solver.train_step(model, ex, opt, sch, step)
torch.cuda.synchronize()
ms = (time.time() - t0) / n * 1e3
print(f"train step: batch {B} x 300000 points, fp32: {ms:.1f} ms/step = {B / ms * 1e3:.2f} samples/s (loss {v:.3f}); "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
