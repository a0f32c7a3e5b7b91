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
        torch._foreach_mul_(self.params, 1 - self.wd * lr)      # a handful of launches for the 570 tensors
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


class GradBuckets:
    """Bucketed gradient all-reduce overlapped with backward (what DistributedDataParallel does for the
    reference, det3d/torchie/apis/train.py:284-290), without the reference's second full all-reduce
    (core/utils/dist_utils.py:51-57, idempotent on already-averaged gradients).

    Parameters are packed, in reverse registration order (the order backward produces gradients), into
    flat fp32 buckets of about `bucket_mb`; every parameter's `.grad` is a view into its bucket, so
    nothing is copied.  A post-accumulate hook counts ready parameters; a full bucket is all-reduced
    asynchronously as soon as all EARLIER buckets have been launched (every rank issues the same
    sequence of collectives whatever order its gradients arrive in).  `finish()` launches what is left
    -- buckets holding parameters that got no gradient this step, find_unused_parameters=True in the
    reference -- waits, and divides by the world size.  xGMI is point-to-point (one ring link carries
    the message), so buckets are few and large: 16 MiB -> 4 messages for PillarNet-18's 59 MB."""

    def __init__(self, params, bucket_mb=16.0, group=None):
        self.group = group
        self.params = [p for p in params if p.requires_grad]
        limit = int(bucket_mb * (1 << 20)) // 4
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            if cur and size + p.numel() > limit:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += p.numel()
        if cur:
            self.buckets.append(cur)
        self.flat, self.bucket_of = [], {}
        for b, ps in enumerate(self.buckets):
            dev = ps[0].device
            flat = torch.zeros(sum(p.numel() for p in ps), dtype=torch.float32, device=dev)
            off = 0
            for p in ps:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self.bucket_of[p] = b
                p.register_post_accumulate_grad_hook(self._on_grad)
            self.flat.append(flat)
        self.launched_early = 0
        self._reset()

    def _reset(self):
        self.ready = [0] * len(self.buckets)
        self.next = 0
        self.handles = []
        self.active = False

    def begin(self, loss=None):
        """Before backward: zero the buckets (gradients accumulate into the views) and arm the hooks.
        With `loss`, parameters the autograd graph of `loss` does not reach are counted as ready at
        once (DDP's find_unused_parameters=True, apis/train.py:289), so they do not hold their bucket
        -- and every later one -- back until finish()."""
        for flat in self.flat:
            flat.zero_()
        self._reset()
        self.active = True
        if loss is not None and loss.grad_fn is not None:
            used, seen, stack = set(), set(), [loss.grad_fn]
            while stack:
                fn = stack.pop()
                if fn is None or fn in seen:
                    continue
                seen.add(fn)
                v = getattr(fn, "variable", None)
                if v is not None:
                    used.add(v)
                stack.extend(f for f, _ in fn.next_functions)
            for p in self.params:
                if p not in used:
                    self.ready[self.bucket_of[p]] += 1
            self._launch_ready()

    def _distributed(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _launch_ready(self, force=False):
        while self.next < len(self.buckets) and (force or self.ready[self.next] == len(self.buckets[self.next])):
            if self._distributed():
                self.handles.append(dist.all_reduce(self.flat[self.next], group=self.group, async_op=True))
                if not force:
                    self.launched_early += 1
            self.next += 1

    def _on_grad(self, p):
        if not self.active:
            return
        self.ready[self.bucket_of[p]] += 1
        self._launch_ready()

    def finish(self):
        """After backward: every bucket reduced and averaged; `.grad` of every parameter is final."""
        self._launch_ready(force=True)
        for h in self.handles:
            h.wait()
        if self._distributed():
            w = float(dist.get_world_size(self.group))
            for flat in self.flat:
                flat.div_(w)
        self.active = False


class DataParallelTrainer:
    """One process per GPU; the step of Trainer.batch_processor_inline + DistOptimizerHook
    (det3d/torchie/trainer/trainer.py:358-389, core/utils/dist_utils.py:45-57) with the wrapping of
    train_detector (apis/train.py:267-311): synchronised BN (apex convert_syncbn_model there), gradient
    buckets overlapped with backward, clip 35, one-cycle true-weight-decay Adam."""

    def __init__(self, model, lr_max=1e-3, wd=0.01, total_steps=100, sync_bn=True, overlap=True,
                 bucket_mb=16.0, grad_clip=35.0, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4):
        self.model = model
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.sync_bn = bool(sync_bn and world > 1)
        if self.sync_bn:
            from .nn_utils import convert_syncbn_model
            convert_syncbn_model(model)
        if world > 1:                     # DDP broadcasts rank 0's parameters and buffers at construction
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, 0)
        self.overlap = bool(overlap)
        self.buckets = GradBuckets(model.parameters(), bucket_mb) if overlap else None
        self.optimizer = TrueWdAdam(model.parameters(), wd=wd)
        self.schedule = OneCycle(total_steps, lr_max, moms, div_factor, pct_start)
        self.grad_clip = grad_clip

    def step(self, example, step):
        self.optimizer.set(*self.schedule.at(step))
        losses = self.model(example, return_loss=True)
        loss = sum(losses["loss"])
        if self.buckets is not None:
            self.buckets.begin(loss)
            loss.backward()
            self.buckets.finish()
        else:
            self.optimizer.zero_grad()
            loss.backward()
            allreduce_grads(self.model.parameters())
        if self.grad_clip:
            torch.nn.utils.clip_grad_norm_([p for p in self.model.parameters() if p.grad is not None],
                                           self.grad_clip)
        self.optimizer.step()
        # the loss stays on the device: float(loss) is the caller's (only) host wait, so a loop that reads it every
        # N steps, as the reference's logger hook does, lets the host queue the next step meanwhile
        return loss.detach(), losses


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
