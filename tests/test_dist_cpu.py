"""world_size-2 gloo rehearsal of the N>1 path (frame sharding, max-over-ranks timing, detection
gather) -- the same helpers bench.py / a dist_test-style driver use over RCCL."""
import os
import socket

import torch
import torch.multiprocessing as mp

from pillarnet_lts_amd import dist_utils


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = dist_utils.init_dist("gloo")
    ids = dist_utils.shard_indices(7, r, w)
    dets = {f"tok{i}": {"scores": torch.full((2,), float(i))} for i in ids}
    merged = dist_utils.gather_detections(dets)
    t = dist_utils.max_over_ranks(1.0 + rank)
    q.put((rank, ids, sorted(merged), t))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 0]   # padded by wrapping
    for _, _, toks, t in res:
        assert toks == [f"tok{i}" for i in range(7)]
        assert t == 2.0


def test_single_process_paths():
    assert dist_utils.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert dist_utils.max_over_ranks(3.5) == 3.5
    assert dist_utils.gather_detections({"a": 1}) == {"a": 1}
