"""Optimiser side of the reference's training loop, restated: one-cycle schedule
(det3d/solver/learning_schedules_fastai.py:70-95), Adam with decoupled ("true") weight decay as
fastai's OptimWrapper applies it (det3d/solver/fastai_optim.py:158-175), gradient clipping and the
data-parallel gradient all-reduce (det3d/core/utils/dist_utils.py:8-57)."""
import math

import torch
import torch.distributed as dist


def annealing_cos(start, end, pct):
    return end + (start - end) / 2 * (math.cos(math.pi * pct) + 1)


class OneCycle:
    """lr: low -> lr_max over pct_start of the run, then -> low/1e4; momentum moves the other way."""

    def __init__(self, total_step, lr_max, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4):
        self.total, self.lr_max, self.moms, self.low = total_step, lr_max, tuple(moms), lr_max / div_factor
        self.a1 = int(total_step * pct_start)

    def at(self, step):
        if step < self.a1:
            p = step / max(self.a1, 1)
            return annealing_cos(self.low, self.lr_max, p), annealing_cos(self.moms[0], self.moms[1], p)
        p = (step - self.a1) / max(self.total - self.a1, 1)
        return annealing_cos(self.lr_max, self.low / 1e4, p), annealing_cos(self.moms[1], self.moms[0], p)


class TrueWdAdam:
    """Adam(betas=(mom, 0.99)) + `p *= 1 - wd*lr` before the step (fixed_wd=True in the configs)."""

    def __init__(self, params, wd=0.01, betas=(0.9, 0.99)):
        self.params = [p for p in params if p.requires_grad]
        self.opt = torch.optim.Adam(self.params, lr=1e-3, betas=betas)
        self.wd = wd

    def set(self, lr, mom):
        for g in self.opt.param_groups:
            g["lr"] = lr
            g["betas"] = (mom, g["betas"][1])

    def zero_grad(self):
        self.opt.zero_grad(set_to_none=True)

    @torch.no_grad()
    def step(self):
        lr = self.opt.param_groups[0]["lr"]
        for p in self.params:
            p.mul_(1 - self.wd * lr)
        self.opt.step()


def allreduce_grads(params, world_size=None):
    """One flat all-reduce of every gradient, then / world_size (dist_utils.py:8-42).  The reference
    runs this on top of DDP's own bucketed all-reduce (averaging twice is idempotent); here it is THE
    gradient exchange: one RCCL call per dtype over xGMI."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world_size = world_size or dist.get_world_size()
    grads = [p.grad for p in params if p.requires_grad and p.grad is not None]
    by_type = {}
    for g in grads:
        by_type.setdefault(g.dtype, []).append(g)
    for bucket in by_type.values():
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat)
        flat.div_(world_size)
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n


def train_step(model, example, optimizer, schedule=None, step=0, grad_clip=35.0):
    """forward(return_loss=True) -> sum of task losses -> backward -> all-reduce -> clip -> step
    (Trainer.batch_processor_inline + DistOptimizerHook.after_train_iter)."""
    if schedule is not None:
        optimizer.set(*schedule.at(step))
    losses = model(example, return_loss=True)
    loss = sum(losses["loss"])
    optimizer.zero_grad()
    loss.backward()
    allreduce_grads(model.parameters())
    if grad_clip:
        torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], grad_clip)
    optimizer.step()
    return float(loss.detach()), losses
