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


# ---- data-parallel training pieces under two gloo ranks ---------------------------------------------
def _spawn(fn, world=2, timeout=180):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=fn, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=timeout) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.body = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16),
                                        torch.nn.ReLU(), torch.nn.Linear(16, 3))
        self.unused = torch.nn.Linear(4, 4)        # parameters that never get a gradient

    def forward(self, x):
        return self.body(x)


def _net():
    torch.manual_seed(5)
    return _Net()


def _rank_batch(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(5 + 3 * rank, 6, generator=g)


def _bucket_worker(rank, world, port, q):
    from pillarnet_lts_amd import solver
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo")
    net = _net()
    buckets = solver.GradBuckets(net.parameters(), bucket_mb=16 * 17 * 4 / (1 << 20))   # ~one layer per bucket
    out = []
    for it in range(2):                          # twice: the second step must not see the first one's sums
        loss = net(_rank_batch(rank) * (it + 1)).pow(2).sum()
        buckets.begin(loss)
        loss.backward()
        buckets.finish()
        out.append([p.grad.clone().numpy() for p in net.parameters()])
    q.put((rank, out, len(buckets.buckets), buckets.launched_early))
    torch.distributed.destroy_process_group()


def test_overlapped_bucketed_allreduce_equals_single_process_mean():
    import numpy as np
    res = _spawn(_bucket_worker)
    assert res[0][2] >= 3 and res[0][3] >= 2       # several buckets, some launched during backward
    for it in range(2):
        want = None
        for rank in range(2):
            net = _net()
            net(_rank_batch(rank) * (it + 1)).pow(2).sum().backward()
            g = [torch.zeros_like(p) if p.grad is None else p.grad for p in net.parameters()]
            want = g if want is None else [a + b for a, b in zip(want, g)]
        want = [w / 2 for w in want]
        for rank in range(2):
            got = res[rank][1][it]
            assert len(got) == len(want)
            for a, b in zip(got, want):
                np.testing.assert_allclose(a, b.numpy(), rtol=1e-5, atol=1e-6)


def _syncbn_worker(rank, world, port, q):
    from pillarnet_lts_amd import nn_utils
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo")
    torch.manual_seed(3)
    holder = torch.nn.Sequential(torch.nn.BatchNorm1d(4, eps=1e-3, momentum=0.01), torch.nn.BatchNorm2d(4, eps=1e-3))
    with torch.no_grad():
        for bn in holder:
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_()
    keys = list(holder.state_dict().keys())
    nn_utils.convert_syncbn_model(holder)
    assert list(holder.state_dict().keys()) == keys and isinstance(holder[0], nn_utils.SyncBatchNorm)
    x1, w1, x2, w2 = _syncbn_inputs(rank)
    x1.requires_grad_(True)
    x2.requires_grad_(True)
    y1, y2 = holder[0](x1), holder[1](x2)
    ((y1 * w1).sum() + (y2 * w2).sum()).backward()
    q.put((rank, y1.detach().numpy(), x1.grad.numpy(), holder[0].weight.grad.numpy(), holder[0].bias.grad.numpy(),
           holder[0].running_mean.numpy(), holder[0].running_var.numpy(), y2.detach().numpy(), x2.grad.numpy(),
           holder[1].weight.grad.numpy()))
    torch.distributed.destroy_process_group()


def _syncbn_inputs(rank):
    g = torch.Generator().manual_seed(40 + rank)
    n = 7 + 5 * rank                                # ranks hold different numbers of rows
    return (torch.randn(n, 4, generator=g) * 2 + 1, torch.randn(n, 4, generator=g),
            torch.randn(1 + rank, 4, 3, 5, generator=g), torch.randn(1 + rank, 4, 3, 5, generator=g))


def test_syncbn_two_ranks_equals_batchnorm_on_the_joined_batch():
    import numpy as np
    res = _spawn(_syncbn_worker)
    torch.manual_seed(3)
    ref = torch.nn.Sequential(torch.nn.BatchNorm1d(4, eps=1e-3, momentum=0.01), torch.nn.BatchNorm2d(4, eps=1e-3))
    with torch.no_grad():
        for bn in ref:
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_()
    ins = [_syncbn_inputs(r) for r in range(2)]
    x1 = torch.cat([i[0] for i in ins]).requires_grad_(True)
    w1 = torch.cat([i[1] for i in ins])
    x2 = torch.cat([i[2] for i in ins]).requires_grad_(True)
    w2 = torch.cat([i[3] for i in ins])
    y1, y2 = ref[0](x1), ref[1](x2)
    ((y1 * w1).sum() + (y2 * w2).sum()).backward()
    n0, b0 = ins[0][0].shape[0], ins[0][2].shape[0]
    rows = [slice(0, n0), slice(n0, None)]
    imgs = [slice(0, b0), slice(b0, None)]
    tol = dict(rtol=1e-4, atol=1e-5)
    for r in range(2):
        _, gy1, gx1, gw, gb, rm, rv, gy2, gx2, gw2 = res[r]
        np.testing.assert_allclose(gy1, y1.detach()[rows[r]].numpy(), **tol)
        np.testing.assert_allclose(gx1, x1.grad[rows[r]].numpy(), **tol)
        np.testing.assert_allclose(gy2, y2.detach()[imgs[r]].numpy(), **tol)
        np.testing.assert_allclose(gx2, x2.grad[imgs[r]].numpy(), **tol)
        np.testing.assert_allclose(rm, ref[0].running_mean.numpy(), **tol)
        np.testing.assert_allclose(rv, ref[0].running_var.numpy(), **tol)
    # parameter gradients are local sums: the ranks' sums add up to the joined batch's gradient
    np.testing.assert_allclose(res[0][3] + res[1][3], ref[0].weight.grad.numpy(), **tol)
    np.testing.assert_allclose(res[0][4] + res[1][4], ref[0].bias.grad.numpy(), **tol)
    np.testing.assert_allclose(res[0][9] + res[1][9], ref[1].weight.grad.numpy(), **tol)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (dry run: gloo, CPU
    step) and rank 0 prints ONE JSON line with n_gpus 2; a failing rank makes the launcher exit non-zero."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup",
                        "1", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["dry_run"] is True
    assert rec["scaling"] == "weak" and rec["value"] is None
    assert rec["ranks_seen"] == 2          # counted by an all-reduce of ones, not read from the environment
    # under an external launcher (WORLD_SIZE set) it does not spawn: one rank of a 1-rank world
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup",
                         "0", "--dry-run", "--mode", "train"], env=env, capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and json.loads(r1.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dtype", "int8"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0
