"""Where a training step's host time goes: per phase (reader, backbone, neck, head, loss, backward, clip + step), wall
time with a device wait at both ends and the number of kernel launches torch's profiler sees."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from pillarnet_lts_amd import presets, solver, synth, targets

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = presets.nusc_pillarnet18()
model = presets.build(cfg).to(dev).train()
B = 4
rng = cfg.model.reader.pc_range
samples = [targets.synthetic_example(synth.nusc_frame(10, 30000, seed=s), cfg.model.bbox_head.tasks, rng,
                                     cfg.pillar_size, n_boxes=40, seed=s, train_cfg=cfg.train_cfg) for s in range(B)]
ex = targets.collate(samples, dev)
opt = solver.TrueWdAdam(model.parameters(), wd=0.01)
sch = solver.OneCycle(100, 1e-3)
for step in range(3):
    solver.train_step(model, ex, opt, sch, step)
torch.cuda.synchronize()


def phase(name, fn):
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        t0 = time.time()
        out = fn()
        t_host = time.time() - t0
        torch.cuda.synchronize()
        t_all = time.time() - t0
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    busy = sum(e.device_time for e in evs) / 1e3
    print(f"{name:12s} wall {t_all * 1e3:7.1f} ms  host queueing {t_host * 1e3:7.1f} ms  device busy {busy:7.1f} ms  launches {len(evs):5d}",
          flush=True)
    return out


data = dict(points=ex["points"], batch_size=len(ex["metadata"]))
sp = phase("reader", lambda: model.reader(data))
feats = phase("backbone", lambda: model.backbone(sp))
bev = phase("neck", lambda: model.neck(feats))
preds = phase("head", lambda: model.bbox_head(bev))
losses = phase("loss", lambda: model.bbox_head.loss(ex, preds, model.train_cfg))
loss = sum(losses["loss"])
opt.zero_grad()
phase("backward", lambda: loss.backward())
phase("clip", lambda: torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 35.0))
phase("step", lambda: opt.step())
