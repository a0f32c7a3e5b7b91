"""Multi-GPU plumbing of the inference path: one process per GPU, frames sharded, no collective on
the data path; detections gathered once at the end.

Mirrors: tools/dist_test.py:86-127,177-178 (init, frame loop, gather), DistributedSampler with
shuffle=False (det3d/datasets/loader/build_loader.py:33-36), and `all_gather(data)` of
det3d/torchie/trainer/utils.py:114-154 (the reference pickles to a ByteTensor by hand;
torch.distributed's object collective does the same job on RCCL and gloo)."""
import os

import torch
import torch.distributed as dist


def init_dist(backend="nccl"):
    """env:// rendezvous as torch.distributed.run sets it up; backend "nccl" is RCCL on ROCm."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_indices(num_frames, rank, world):
    """Frame ids of this rank: strided, padded by wrapping so every rank gets the same count
    (DistributedSampler(shuffle=False) semantics)."""
    per = (num_frames + world - 1) // world
    total = per * world
    ids = list(range(num_frames)) + list(range(total - num_frames))
    return ids[rank:total:world]


def max_over_ranks(seconds, device=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_detections(local_dict):
    """{token: detection} of every rank merged on all ranks (dist_test.py:177-187)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(local_dict)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local_dict)
    merged = {}
    for p in parts:
        merged.update(p)
    return merged
