#!/usr/bin/env python3
"""Headline benchmark: frames/s of nuScenes 10-sweep PillarNet-18 inference (BASELINE.json
configs[1]) through the det3d-API detector on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one synthetic frame (10 sweeps x 30 000 points, resident in HBM) through
`detector(example, return_loss=False)`: voxelize -> PFN -> sparse backbone -> neck -> CenterHead ->
decode -> rotated NMS -> detections, batch 1 per GPU as tools/dist_test.py --speed_test does
(dist_test.py:111,142-155).  Frames are independent, so N GPUs run N frame streams with no
data-path collective ("weak" scaling); the timed region is bracketed by barrier + synchronize and
the rank maximum is reported.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--probe", default="256x256", help="conv shape cinxcout for the roofline leg")
    return ap.parse_args()


def cpu_baseline(model, cfg):
    """The oracle (CPU restatement of the reference forward) timed on this host: ONE frame of
    configs[0] (1 sweep, 30k points, fp32), all host cores for the torch-CPU dense part."""
    import copy
    from oracle import detector_oracle as do
    from pillarnet_lts_amd import presets, synth
    from pillarnet_lts_amd.center_utils import set_by_task_cfg
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 32)   # torch-CPU conv stops scaling long before 256 threads
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    test_cfg = set_by_task_cfg(copy.deepcopy(presets.nusc_pillarnet18().test_cfg), model.bbox_head.num_classes)
    pts = [synth.nusc_sweep(30000, seed=0)]
    t0 = time.time()
    dets = do.pillarnet_forward(sd, pts, cfg.model, test_cfg)
    dt = time.time() - t0
    return {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"1 frame of configs[0] (1 sweep, 30000 points, fp32, {len(dets[0]['scores'])} detections) "
                      f"through oracle/detector_oracle.py in {dt:.1f} s"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from pillarnet_lts_amd import presets, sparse_conv, synth
    torch.manual_seed(0)
    cfg = presets.nusc_pillarnet18()
    model = presets.build(cfg)
    synth.randomize_bn_(model, seed=1)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model_cpu_sd = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        model_cpu_sd = True
        cpu = cpu_baseline(model, cfg)
    model.to_inference(dtype)

    # a few distinct frames per rank, all resident in HBM before the clock starts
    frames = [torch.from_numpy(synth.nusc_frame(10, 30000, seed=100 * rank + i)).to(dev) for i in range(4)]
    examples = [{"points": [f], "metadata": [{"token": i}]} for i, f in enumerate(frames)]

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for i in range(args.warmup):
            model(examples[i % len(examples)], return_loss=False)
        cin, cout = (int(v) for v in args.probe.split("x"))
        sparse_conv.EVENT_PROBE.update(shape=(cin, cout), events=[])
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = model(examples[i % len(examples)], return_loss=False)
        barrier()
        t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # roofline of the dominant hand-written kernel (sparse implicit GEMM at the probed width):
    # algorithmic flops = 2*Cin*Cout * (active (site, offset) pairs), from the live rulebook
    ev = sparse_conv.EVENT_PROBE["events"]
    roof = None
    if ev:
        avg_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        nbr, rows = sparse_conv.EVENT_PROBE["rules"], int(sparse_conv.EVENT_PROBE["rows"].item())
        pairs = int((nbr[:, :rows] >= 0).sum().item())
        flops = 2.0 * cin * cout * pairs
        peak = 2500.0 if dtype == torch.bfloat16 else 157.3
        ach = flops / (avg_ms * 1e-3) / 1e12
        traffic = None   # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/, same shape)
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pmc = json.load(f)
                traffic = (pmc.get(f"spconv_{cin}x{cout}_window") or pmc.get(f"spconv_{cin}x{cout}", {})).get("hbm_bytes_per_launch")
        except OSError:
            pass
        roof = {"bound": "mfma", "kernel": f"gg_win_kernel<128,256,2,4,8,64,192,{args.dtype}> (window-staged gather-GEMM, sparse SubM {cin}->{cout})",
                "achieved": round(ach, 2),
                "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                "launches": len(ev), "avg_us": round(avg_ms * 1e3, 2), "active_pairs": pairs, "rows": rows}
    sparse_conv.EVENT_PROBE.update(shape=None, events=[], rules=None, rows=None)

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
                       "detections_last_frame": int(out[0]["scores"].shape[0])},
            "roofline": roof,
            "cpu_baseline": cpu if model_cpu_sd else None,
        }
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
