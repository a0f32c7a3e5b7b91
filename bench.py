#!/usr/bin/env python3
"""Headline benchmark: frames/s of nuScenes 10-sweep PillarNet-18 inference (BASELINE.json
configs[1]) through the det3d-API detector on MI355X.

    python bench.py --gpus N --steps K --warmup W            # starts N ranks itself when N > 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --mode train --gpus N                    # configs[3]: data-parallel train step

A step = one synthetic frame (10 sweeps x 30 000 points, resident in HBM) through
`detector(example, return_loss=False)`: voxelize -> PFN -> sparse backbone -> neck -> CenterHead ->
decode -> rotated NMS -> detections, batch 1 per GPU as tools/dist_test.py --speed_test does
(dist_test.py:111,142-155).  Frames are independent, so N GPUs run N frame streams with no
data-path collective ("weak" scaling), one process per GPU as tools/dist_test.py:86-127 is
launched; the timed region is bracketed by barrier + synchronize and the rank maximum is reported.
Prints ONE JSON line on rank 0.

When WORLD_SIZE is not in the environment and --gpus N > 1, this process only launches: it starts
N children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set), never
touches the GPU itself and exits non-zero if any child fails.

--dry-run replaces the GPU frame by a trivial CPU step over the gloo backend: it exists so that the
launcher, rendezvous, barrier, rank-maximum and JSON contract can be tested without a GPU, and
marks its line `"dry_run": true`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", default="infer", choices=["infer", "train"])
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=2, help="frames in flight per GPU, one stream each")
    ap.add_argument("--depth", type=int, default=None, help="frames queued before the host waits for the oldest (default: lanes)")
    ap.add_argument("--no-shared-cu-tiles", action="store_true", help="record the graphs with the one-launch-owns-the-chip tiles")
    ap.add_argument("--no-graph", action="store_true", help="queue every kernel from Python instead of replaying the frame's HIP graph")
    ap.add_argument("--cpu-baseline-1t", action="store_true",
                    help="also time the whole oracle frame on ONE thread (minutes)")
    ap.add_argument("--probe", default="256x256", help="conv shape cinxcout for the roofline leg")
    ap.add_argument("--dry-run", action="store_true", help="gloo + CPU no-op step (launcher test)")
    a = ap.parse_args(argv)
    if a.steps is None:
        a.steps = 200 if a.mode == "infer" else 5     # 200 frames = 0.25 s: the two-frame pipeline's fill and drain are < 1 %
    if a.warmup is None:
        a.warmup = 8 if a.mode == "infer" else 2
    if a.dtype is None:
        a.dtype = "bf16" if a.mode == "infer" else "fp32"   # the reference trains in fp32
    return a


# ---- launcher (parent process: no GPU call, no re-exec) ----------------------------------------
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(args, argv):
    """One child per GPU (tools/dist_test.py is started the same way by torch.distributed.launch:
    dist_test.py:86-127).  Rank 0's stdout is this process's stdout."""
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(len(procs)))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in pending:
                    procs[q].terminate()          # exact children of this process only
        time.sleep(0.05)
    return rc


# ---- CPU baseline (rank 0, N = 1) ----------------------------------------------------------------
def cpu_baseline(model, cfg, full_1t=False):
    """The oracle (this repository's CPU restatement of the reference forward, kind "port") timed
    on the host: ONE frame of the SAME workload as the GPU line (configs[1]: 10 sweeps x 30 000
    points, fp32).  Also timed: the voxelize stage alone on one thread -- the reference's hard
    voxelizer `points_to_voxel` (the "det3d CPU voxelizer" of configs[0], caps of
    configs/nusc/voxelnet/nusc_centerpoint_pillarnet_h.py:151-158) and the dynamic pillarizer +
    PFN + scatter-max."""
    import copy
    import torch
    from oracle import detector_oracle as do
    from oracle import pillar_oracle, voxelize
    from pillarnet_lts_amd import presets, synth
    from pillarnet_lts_amd.center_utils import set_by_task_cfg
    host = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(host, 32)   # torch-CPU convolutions stop scaling (and then slow down) beyond ~32 threads
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    test_cfg = set_by_task_cfg(copy.deepcopy(presets.nusc_pillarnet18().test_cfg), model.bbox_head.num_classes)
    frame = synth.nusc_frame(10, 30000, seed=0)
    rd = cfg.model.reader

    def whole(threads):
        torch.set_num_threads(threads)
        t0 = time.time()
        dets = do.pillarnet_forward(sd, [frame], cfg.model, test_cfg)
        return time.time() - t0, len(dets[0]["scores"])

    # voxelize stage, one thread (numpy / C are single-threaded here)
    torch.set_num_threads(1)
    t0 = time.time()
    vox, coors, num = voxelize.points_to_voxel(frame, [0.075, 0.075, 0.2], rd.pc_range, 10, True, 120000)
    t_hard = time.time() - t0
    p = "reader.pfn_layers.shared_mlps"
    g = lambda k: sd[k].numpy()
    t0 = time.time()
    r = pillar_oracle.pillarize([frame], rd.pc_range, rd.pillar_size, g(p + ".0.weight"), g(p + ".1.weight"),
                                g(p + ".1.bias"), g(p + ".1.running_mean"), g(p + ".1.running_var"), 1e-3)
    t_dyn = time.time() - t0
    dt, ndet = whole(cores)
    out = {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": cores, "host_cores": host, "kind": "port",
           "sample": f"1 frame of configs[1] (10 sweeps x 30000 points, fp32, {ndet} detections) through "
                     f"oracle/detector_oracle.py in {dt:.1f} s on {cores} threads",
           "voxelize_one_thread": {
               "hard_voxelizer_s": round(t_hard, 4), "hard_voxels": int(vox.shape[0]),
               "dynamic_pillarize_pfn_scatter_s": round(t_dyn, 4), "pillars": int(r["indices"].shape[0]),
               "note": "points_to_voxel restatement (voxel 0.075x0.075x0.2, 10 points/voxel, 120000 voxels) and "
                       "the dynamic reader, 300000 points, 1 thread"}}
    # configs[0]'s own workload: ONE sweep of 30 000 points through the CPU voxelizer and the CPU forward (the GPU side
    # of that frame is pinned by tests/test_fullsize_gpu.py's nusc1 cases, not timed here)
    frame1 = synth.nusc_frame(1, 30000, seed=0)
    torch.set_num_threads(1)
    t0 = time.time()
    vox1, _, _ = voxelize.points_to_voxel(frame1, [0.075, 0.075, 0.2], rd.pc_range, 10, True, 120000)
    t_hard1 = time.time() - t0
    torch.set_num_threads(cores)
    t0 = time.time()
    d1 = do.pillarnet_forward(sd, [frame1], cfg.model, test_cfg)
    dt0 = time.time() - t0
    out["configs0"] = {"value": round(1.0 / (dt0 + t_hard1), 4), "unit": "frames/s", "cores": cores, "kind": "port",
                       "sample": f"1 frame of configs[0] (1 sweep x 30000 points, fp32, {len(d1[0]['scores'])} detections): hard "
                                 f"voxelizer {t_hard1 * 1e3:.1f} ms on 1 thread ({int(vox1.shape[0])} voxels) + "
                                 f"oracle/detector_oracle.py forward {dt0:.1f} s on {cores} threads"}
    if full_1t:
        dt1, _ = whole(1)
        out["one_thread"] = {"value": round(1.0 / dt1, 5), "unit": "frames/s", "seconds": round(dt1, 1)}
    torch.set_num_threads(cores)
    return out


# ---- rank body -----------------------------------------------------------------------------------
def _dist_setup(args):
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return world, rank, local, dist


def _ranks_seen(dist, dev):
    """How many ranks the collective library actually connects: an all-reduce of ones (RCCL on the GPU, gloo in the
    dry run), not the environment's WORLD_SIZE."""
    import torch
    if dist is None:
        return 1
    t = torch.ones(1, dtype=torch.float32, device=dev if dev is not None else "cpu")
    dist.all_reduce(t)
    return int(round(float(t.item())))


def _min_max_over_ranks(dist, value, dev):
    import torch
    if dist is None:
        return value, value
    lo = torch.tensor([value], dtype=torch.float64, device=dev if dev is not None else "cpu")
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return float(lo.item()), float(hi.item())


def _max_over_ranks(dist, seconds, dev):
    import torch
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def run_dry(args):
    import torch
    world, rank, local, dist = _dist_setup(args)
    x = torch.ones(1024)
    barrier = (lambda: dist.barrier()) if dist is not None else (lambda: None)
    for _ in range(args.warmup):
        x = x * 1.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = x * 1.0
    barrier()
    elapsed = _max_over_ranks(dist, time.perf_counter() - t0, None)
    ranks_seen = _ranks_seen(dist, None)
    if rank == 0:
        print(json.dumps({"metric": "frames/sec, nuScenes 10-sweep PillarNet-18 inference", "dry_run": True,
                          "value": None, "unit": "frames/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 6),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
                          "data": "none (launcher dry run on CPU, gloo)", "mode": args.mode,
                          "config": {"workload": "dry run: no detector step executed"}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def run_infer(args):
    import torch
    world, rank, local, dist = _dist_setup(args)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from pillarnet_lts_amd import presets, sparse_conv, synth
    torch.manual_seed(0)
    cfg = presets.nusc_pillarnet18()
    model = presets.build(cfg)
    synth.randomize_bn_(model, seed=1)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model, cfg, args.cpu_baseline_1t)
    model.to_inference(dtype)

    # a few distinct frames per rank, all resident in HBM before the clock starts
    host_frames = [torch.from_numpy(synth.nusc_frame(10, 30000, seed=100 * rank + i)).pin_memory() for i in range(4)]
    frames = [f.to(dev) for f in host_frames]
    examples = [{"points": [f], "metadata": [{"token": i}]} for i, f in enumerate(frames)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        from pillarnet_lts_amd import FrameGraphs
        args.no_graph = args.no_graph or dtype != torch.bfloat16
        runner = model if args.no_graph else FrameGraphs(model, lanes=args.lanes, shared_cu_tiles=False if args.no_shared_cu_tiles else None)     # one HIP graph per padded frame shape
        for i in range(args.warmup):
            model(examples[i % len(examples)], return_loss=False)
        try:
            for i in range(max(args.warmup, 1)):
                for _ in range(1 if args.no_graph else args.lanes):
                    runner.forward_deferred(examples[i % len(examples)]).result()
        except RuntimeError as e:      # a runtime that cannot capture: same frames, every kernel queued from Python
            if args.no_graph:
                raise
            print(f"bench.py: frame graph capture failed ({e}); running the eager loop", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            args.no_graph, runner = True, model
        # The timed job: every frame's detections reach the host inside the region.  Frame i+1's kernels are queued
        # before the host waits for frame i's counts, so the GPU works through the host's hand-over.
        barrier()
        t0 = time.perf_counter()
        depth = args.depth if args.depth is not None else (1 if args.no_graph else args.lanes)
        inflight = []
        for i in range(args.steps):
            inflight.append(runner.forward_deferred(examples[i % len(examples)]))
            if len(inflight) > depth:
                out = inflight.pop(0).result()
        for h in inflight:
            out = h.result()
        own = time.perf_counter() - t0            # this rank's own stream of frames (before the closing barrier)
        barrier()
        elapsed = _max_over_ranks(dist, time.perf_counter() - t0, dev)
        rank_fps = _min_max_over_ranks(dist, args.steps / own, dev)
        ranks_seen = _ranks_seen(dist, dev)

        # the same frames with the host waiting after each one (the shape of dist_test.py's loop): every kernel queued
        # from Python, and -- what a caller of this package would run in that loop -- one HIP-graph replay per frame on
        # one stream (tiles of the launch-owns-the-chip policy), the host taking each frame's detections before the next
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = model(examples[i % len(examples)], return_loss=False)
        barrier()
        elapsed_sync_eager = _max_over_ranks(dist, time.perf_counter() - t0, dev)
        elapsed_sync = elapsed_sync_eager
        if not args.no_graph:
            single = FrameGraphs(model, lanes=1)
            for i in range(len(examples)):
                single(examples[i])
            barrier()
            t0 = time.perf_counter()
            for i in range(args.steps):
                out = single(examples[i % len(examples)])
            barrier()
            elapsed_sync = _max_over_ranks(dist, time.perf_counter() - t0, dev)
            del single

        # ---- after the timed region: PCIe-inclusive rate, roofline and reader probes ------------
        barrier()
        t0 = time.perf_counter()
        inflight = []
        for i in range(args.steps):
            pts = host_frames[i % len(host_frames)].to(dev, non_blocking=True)
            inflight.append(runner.forward_deferred({"points": [pts], "metadata": [{"token": i}]}))
            if len(inflight) > depth:
                inflight.pop(0).result()
        for h in inflight:
            h.result()
        barrier()
        elapsed_h2d = _max_over_ranks(dist, time.perf_counter() - t0, dev)

        cin, cout = (int(v) for v in args.probe.split("x"))
        peak = 2500.0 if dtype == torch.bfloat16 else 157.3

        def probe_conv(policy, busy_lane):
            """HIP-event pairs on the launch stream around every sparse `cin -> cout` launch of eager frames recorded
            under tile policy `policy`; with `busy_lane` the other lane's frame graph is replayed beside them on its own
            stream, as in the timed loop.  -> (average ms, launches, active (site, tap) pairs, rows, row capacity the
            launch is sized by)."""
            from pillarnet_lts_amd import _lib
            old_policy = _lib.lib().pn_gg_set_tile_policy(policy)
            sparse_conv.EVENT_PROBE.update(shape=(cin, cout), events=[])
            try:
                for rep in range(4):
                    if busy_lane is not None:
                        fr = next(iter(busy_lane.frames.values()))
                        with torch.cuda.stream(busy_lane.stream):
                            for _ in range(6):               # ~6 ms of the other frame's kernels queued ahead
                                fr.graph.replay()
                    for i in range(2):
                        model(examples[(2 * rep + i) % len(examples)], return_loss=False)
                    torch.cuda.synchronize()
            finally:
                _lib.lib().pn_gg_set_tile_policy(old_policy)
            ev = sparse_conv.EVENT_PROBE["events"]
            if not ev:
                return None
            avg_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
            nbr, rows = sparse_conv.EVENT_PROBE["rules"], int(sparse_conv.EVENT_PROBE["rows"].item())
            pairs = int((nbr[:, :rows] >= 0).sum().item())
            sparse_conv.EVENT_PROBE.update(shape=None, events=[], rules=None, rows=None)
            return avg_ms, len(ev), pairs, rows, int(nbr.shape[1])

        def pmc_traffic(key, *names):
            # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/, same layer shape)
            for name in names:
                try:
                    with open(os.path.join(ROOT, "profiles", name)) as f:
                        hit = json.load(f).get(key)
                except OSError:
                    continue
                if hit:
                    return hit.get("hbm_bytes_per_launch")
            return None

        # (1) the kernel the TIMED loop runs for this layer: graphs are recorded under the shared-CU tile policy, where a
        # 256-wide stride-1 layer is 256 x 256 tiles on fused waves -- half as many workgroups as CUs, the other frame's
        # kernels beside it.  Measured the same way: policy 1, the other lane replaying its graph.
        roof = roof_one = None
        shared = (not args.no_graph) and getattr(runner, "shared_cu_tiles", False)
        if shared and len(runner.lanes) > 1:
            alone = probe_conv(1, None)          # the same launches with the other lane idle: what a kernel trace, which
            got = probe_conv(1, runner.lanes[1])  # serialises the streams, reports for this kernel
            if got:
                avg_ms, n, pairs, rows, cap = got
                ach = 2.0 * cin * cout * pairs / (avg_ms * 1e-3) / 1e12
                tiles = -(-cap // 256) * -(-cout // 256)
                cus = min(256, tiles)
                roof = {"bound": "mfma",
                        "kernel": f"gg_win_kernel<256,256,4,2,0,64,264,{args.dtype},...,FUSED> (window-staged gather-GEMM on 256x256 "
                                  f"tiles, sparse SubM {cin}->{cout}; the kernel the timed graphs run for the 256-wide layers)",
                        "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                        "traffic": pmc_traffic(f"spconv_{cin}x{cout}_fused", "r03_pmc_traffic.json"), "launches": n, "avg_us": round(avg_ms * 1e3, 2),
                        "avg_us_other_lane_idle": round(alone[0] * 1e3, 2) if alone else None,
                        "active_pairs": pairs, "rows": rows, "workgroups": tiles, "cus_used": cus,
                        "frac_of_occupied_cus": round(ach / (peak * cus / 256.0), 4),
                        "measured": "HIP events around the launches of eager frames under the graphs' tile policy while "
                                    "the other lane replays its frame graph"}
        # (2) the same layer when a launch owns the chip (one frame at a time): 128 x 256 tiles, one per CU
        got = probe_conv(0, None)
        if got:
            avg_ms, n, pairs, rows, cap = got
            ach = 2.0 * cin * cout * pairs / (avg_ms * 1e-3) / 1e12
            roof_one = {"bound": "mfma", "kernel": f"gg_win_kernel<128,256,2,4,8,...,{args.dtype}> (one launch owns the chip: "
                                                   f"one-frame-at-a-time loop), sparse SubM {cin}->{cout}",
                        "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                        "traffic": pmc_traffic(f"spconv_{cin}x{cout}_window", "r03_pmc_traffic.json", "r02_pmc_traffic.json"), "launches": n,
                        "avg_us": round(avg_ms * 1e3, 2), "active_pairs": pairs, "rows": rows}
        if roof is None:
            roof = roof_one

        # (3) the frame's matrix work, counted on the live rulebooks of one eager frame: what this path issues (ranked
        # regression heads: towers only at the selected cells), and the share of the bf16 MFMA peak the timed loop reaches
        from pillarnet_lts_amd import _lib
        _lib.FLOP_LOG = []
        model(examples[0], return_loss=False)
        torch.cuda.synchronize()
        by_tag = {}
        for tag, per_pair, pairs_ in _lib.FLOP_LOG:
            by_tag[tag] = by_tag.get(tag, 0.0) + per_pair * float(pairs_)
        _lib.FLOP_LOG = None
        frame_flops = sum(by_tag.values())
        top = sorted(by_tag.items(), key=lambda kv: -kv[1])[:8]

        # reader (voxelize + PFN + scatter-max) as an HBM stage: HIP events on the launch stream
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sp = model.reader(examples[0])
        torch.cuda.synchronize()
        reader_graph = None
        if not args.no_graph:                     # replayed from a graph: device time, not the host's queueing time
            try:
                reader_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(reader_graph, capture_error_mode="thread_local"):
                    sp = model.reader(examples[0])
                reader_graph.replay()
                torch.cuda.synchronize()
            except RuntimeError:
                reader_graph = None
                torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            if reader_graph is not None:
                reader_graph.replay()
            else:
                sp = model.reader(examples[0])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        n_pts, n_pil = int(frames[0].shape[0]), int(sp.num_active_host())
        rbytes = 20 * n_pts + 12 * n_pil + 4 * 32 * n_pil
        hbm_stage = {"stage": "reader: pn_pillarize (cell ids, occupancy index, PFN MLP, scatter-max)",
                     "bytes": rbytes, "us": round(us, 2), "GB_per_s": round(rbytes / us / 1e3, 1),
                     "frac_of_8TBps": round(rbytes / us / 1e3 / 8000.0, 4), "points": n_pts, "pillars": n_pil,
                     "note": "algorithmic bytes = 20 N + 12 M + 128 M; time is the whole call (five launches"
                             + (", replayed from a HIP graph)" if reader_graph is not None else ", queued from Python)")}

    if rank == 0:
        frames_total = args.steps * world
        line = {
            "metric": "frames/sec, nuScenes 10-sweep PillarNet-18 inference",
            "value": round(frames_total / elapsed, 3), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: PillarNet-18 nuScenes 10-sweep inference, "
                                   "300000 points/frame, 1440x1440 pillars, batch 1 per GPU, "
                                   "voxelize->PFN->spconv->RPNV1->CenterHead->rotated NMS",
                       "weights": "random init (reference init rules), BN stats randomised",
                       "parallelism": f"{world} independent frame streams (no data-path collective)",
                       "frames_in_flight_per_gpu": depth,
                       "detections_last_frame": int(out[0]["scores"].shape[0])},
            "loop": (f"one HIP graph replay per frame, {args.lanes} frames in flight on {args.lanes} streams; " if not args.no_graph else "") +
                    "frames queued ahead of the host's wait for the oldest frame's detections (forward_deferred)",
            "ranks_seen": ranks_seen,
            "per_rank_frames_per_s": {"min": round(rank_fps[0], 3), "max": round(rank_fps[1], 3)},
            "value_host_wait_per_frame": round(frames_total / elapsed_sync, 3),
            "value_host_wait_per_frame_eager": round(frames_total / elapsed_sync_eager, 3),
            "value_h2d_inclusive": round(frames_total / elapsed_h2d, 3),
            "roofline": roof,
            "roofline_one_lane": roof_one,
            "frame_flops": round(frame_flops),
            "frame_mfma_frac": round(frame_flops / (elapsed / args.steps * world) / (peak * 1e12), 4),
            "frame_flops_top_layers_GF": {k: round(v / 1e9, 2) for k, v in top},
            "hbm_stage": hbm_stage,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def run_train(args):
    """configs[3]: one data-parallel training step of nuScenes PillarNet-18, batch 4 per GPU
    (samples_per_gpu=4, configs/pillarnet/pillarnet_centerhead_nusc.py:160), fp32 as the reference
    trains: forward with losses -> backward -> gradient all-reduce over RCCL -> clip -> one-cycle
    AdamW step (tools/train.py -> torchie/apis/train.py:251-326, core/utils/dist_utils.py:45-57)."""
    import torch
    world, rank, local, dist = _dist_setup(args)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from pillarnet_lts_amd import presets, solver, synth, targets
    torch.manual_seed(0)
    cfg = presets.nusc_pillarnet18()
    model = presets.build(cfg).to(dev).train()
    per_gpu = 4
    rng = cfg.model.reader.pc_range
    samples = [targets.synthetic_example(synth.nusc_frame(10, 30000, seed=200 + 10 * rank + s),
                                         cfg.model.bbox_head.tasks, rng, cfg.pillar_size, n_boxes=40,
                                         seed=200 + 10 * rank + s, train_cfg=cfg.train_cfg) for s in range(per_gpu)]
    ex = targets.collate(samples, dev)
    trainer = solver.DataParallelTrainer(model, wd=0.01, total_steps=max(100, args.steps + args.warmup))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from pillarnet_lts_amd import nn_utils
    for step in range(args.warmup):
        trainer.step(ex, step)
    barrier()
    torch.cuda.reset_peak_memory_stats()
    early0 = trainer.buckets.launched_early if trainer.buckets is not None else 0
    bn0 = nn_utils.SYNC_BN_COLLECTIVES[0]
    t0 = time.perf_counter()
    for step in range(args.warmup, args.warmup + args.steps):
        loss, _ = trainer.step(ex, step)
    own = time.perf_counter() - t0
    barrier()
    elapsed = _max_over_ranks(dist, time.perf_counter() - t0, dev)
    rank_sps = _min_max_over_ranks(dist, args.steps * per_gpu / own, dev)
    ranks_seen = _ranks_seen(dist, dev)
    n_buckets = len(trainer.buckets.buckets) if trainer.buckets is not None else 0
    early = (trainer.buckets.launched_early - early0) / max(args.steps, 1) if trainer.buckets is not None else 0
    bn_coll = (nn_utils.SYNC_BN_COLLECTIVES[0] - bn0) / max(args.steps, 1)

    # matrix work of the step, counted on the live rulebooks of one sample: every convolution (sparse levels, conv5,
    # neck, the DENSE head -- training evaluates every tower at every cell) once forward, once for the data gradient,
    # once for the weight gradient; against the fp32 matrix peak (v_mfma_f32_32x32x2_f32: 157.3 TFLOP/s)
    step_flops = None
    if rank == 0:
        try:
            import copy
            from pillarnet_lts_amd import _lib
            probe = copy.deepcopy(model).to_inference(torch.bfloat16)
            with torch.no_grad():
                _lib.FLOP_LOG = []
                bev, _ = probe.extract_feat(dict(points=[ex["points"][0]], batch_size=1))
                conv = sum(per_pair * float(pairs) for _, per_pair, pairs in _lib.FLOP_LOG)
                _lib.FLOP_LOG = None
            head = 0.0
            for f, sc in zip(bev, probe.bbox_head.share_convs):
                cells = f.shape[2] * f.shape[3]
                head += 2.0 * 9 * sc[0].in_channels * sc[0].out_channels * cells
            for t, th in enumerate(probe.bbox_head.task_heads):
                f = bev[probe.bbox_head.task_idx[t]]
                cells = f.shape[2] * f.shape[3]
                for m in th.modules():
                    if isinstance(m, torch.nn.Conv2d):
                        head += 2.0 * 9 * m.in_channels * m.out_channels * cells
            step_flops = 3.0 * (conv + head) * per_gpu
            del probe
        except Exception as e:      # the accounting must never cost the measurement
            print(f"bench.py: flop accounting skipped ({e})", file=sys.stderr)
            _lib.FLOP_LOG = None

    # the gradient exchange alone: one flat all-reduce of the same byte count, bus bandwidth per the
    # ring formula 2 (N-1)/N * bytes / t against one xGMI link (~153 GB/s)
    n_param = sum(p.numel() for p in model.parameters() if p.requires_grad)
    allreduce = None
    if dist is not None:
        flat = torch.zeros(n_param, dtype=torch.float32, device=dev)
        for _ in range(3):
            dist.all_reduce(flat)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(flat)
        torch.cuda.synchronize()
        t = _max_over_ranks(dist, (time.perf_counter() - t0) / 10, dev)
        bus = 2.0 * (world - 1) / world * flat.numel() * 4 / t / 1e9
        allreduce = {"bytes": flat.numel() * 4, "ms": round(t * 1e3, 3), "bus_GB_per_s": round(bus, 1),
                     "frac_of_xgmi_link_153GBps": round(bus / 153.0, 3)}
    if rank == 0:
        total = args.steps * per_gpu * world
        print(json.dumps({
            "metric": "samples/sec, nuScenes PillarNet-18 data-parallel training step",
            "value": round(total / elapsed, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: PillarNet-18 nuScenes training step, batch 4 per GPU x "
                                   "300000 points, 1440x1440 pillars, losses + backward + gradient all-reduce + "
                                   "clip + one-cycle AdamW", "global_batch": per_gpu * world,
                       "parallelism": f"dp{world}", "sync_bn": trainer.sync_bn, "overlap": trainer.overlap,
                       "loss_last_step": round(float(loss), 4),
                       "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)},
            "step_flops_per_gpu": None if step_flops is None else round(step_flops),
            "fp32_matrix_frac": None if step_flops is None else round(step_flops / (elapsed / args.steps) / 157.3e12, 4),
            "ranks_seen": ranks_seen,
            "per_rank_samples_per_s": {"min": round(rank_sps[0], 3), "max": round(rank_sps[1], 3)},
            "gradient_buckets": {"count": n_buckets, "launched_early_per_step": round(early, 2),
                                 "note": "buckets all-reduced while backward was still running (the rest at finish())"},
            "sync_bn_collectives_per_step": round(bn_coll, 1),
            "allreduce": allreduce}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args, argv))
    if args.dry_run:
        run_dry(args)
    elif args.mode == "train":
        run_train(args)
    else:
        run_infer(args)


if __name__ == "__main__":
    main()
