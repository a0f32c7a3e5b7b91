"""forward_deferred: the postponed host sync returns what the synchronous call returns."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_forward_deferred_matches_forward():
    from pillarnet_lts_amd import presets, synth
    torch.manual_seed(0)
    cfg = presets.nusc_pillarnet18()
    model = presets.build(cfg)
    synth.randomize_bn_(model, seed=1)
    model.to_inference(torch.bfloat16)
    frames = [torch.from_numpy(synth.nusc_frame(10, 6000, seed=s)).cuda() for s in (3, 4, 5)]
    examples = [{"points": [f], "metadata": [{"token": i}]} for i, f in enumerate(frames)]
    with torch.no_grad():
        want = [model(e, return_loss=False) for e in examples]
        # all three frames queued before the first result is taken
        handles = [model.forward_deferred(e) for e in examples]
        got = [h.result() for h in handles]
    for w, g in zip(want, got):
        assert len(w) == len(g) == 1
        assert g[0]["metadata"] == w[0]["metadata"]
        assert w[0]["scores"].shape[0] > 0
        for k in ("box3d_lidar", "scores", "label_preds"):
            assert torch.equal(w[0][k], g[0][k]), k
    assert handles[0].result() is got[0]        # idempotent


def _same(want, got):
    assert len(want) == len(got)
    for w, g in zip(want, got):
        assert g["metadata"] == w["metadata"]
        for k in ("box3d_lidar", "scores", "label_preds"):
            assert torch.equal(w[k], g[k]), k


@pytest.mark.parametrize("lanes", [1, 2])
def test_frame_graph_replays_match_eager(lanes):
    """a captured frame replayed on other frames of the same padded shape, with eager work between the replays"""
    from pillarnet_lts_amd import FrameGraphs, presets, synth
    torch.manual_seed(0)
    cfg = presets.nusc_pillarnet18()
    model = presets.build(cfg)
    synth.randomize_bn_(model, seed=1)
    model.to_inference(torch.bfloat16)
    graphs = FrameGraphs(model, bucket=8192, lanes=lanes)
    pts = [torch.from_numpy(synth.nusc_frame(10, 6000, seed=s)).cuda() for s in (11, 12, 13)]
    # ragged: 60000 points, then fewer (tail of the input buffer goes back to filler), then a full bucket multiple
    frames = [pts[0], pts[1][:50011], pts[2][:57344], pts[0][:49153], pts[1]]
    with torch.no_grad():
        for i, f in enumerate(frames):
            ex = {"points": [f], "metadata": [{"token": i}]}
            want = model(ex, return_loss=False)
            assert want[0]["scores"].shape[0] > 0
            _same(want, graphs(ex))
        assert sorted({k for ln in graphs.lanes for k in ln.frames}) == [(5, (57344,)), (5, (65536,))]
        # two frames in flight, batch of two ragged samples
        ex2 = [{"points": [pts[0][:30000], pts[1][:41000]], "metadata": [{"token": "a"}, {"token": "b"}]},
               {"points": [pts[2][:31000], pts[0][:40000]], "metadata": [{"token": "c"}, {"token": "d"}]}]
        want = [model(e, return_loss=False) for e in ex2]
        for rep in range(3):
            handles = [graphs.forward_deferred(e) for e in ex2]
            for w, h in zip(want, handles):
                _same(w, h.result())


def test_bench_shape_two_lane_graphs_match_eager():
    """The exact shape bench.py times: BASELINE configs[1] frames (10 sweeps x 30 000 points, 1440^2 grid), bf16,
    FrameGraphs(lanes=2) under its shared-CU tile policy, four distinct frames cycled with two in flight (a frame is
    queued before the host takes the oldest one's detections) -- against the eager synchronous call, bit for bit."""
    from pillarnet_lts_amd import FrameGraphs, presets, synth
    torch.manual_seed(0)
    cfg = presets.nusc_pillarnet18()
    model = presets.build(cfg)
    synth.randomize_bn_(model, seed=1)
    model.to_inference(torch.bfloat16)
    frames = [torch.from_numpy(synth.nusc_frame(10, 30000, seed=i)).cuda() for i in range(4)]
    examples = [{"points": [f], "metadata": [{"token": i}]} for i, f in enumerate(frames)]
    with torch.no_grad():
        want = [model(e, return_loss=False) for e in examples]
        assert all(w[0]["scores"].shape[0] > 100 for w in want)
        runner = FrameGraphs(model, lanes=2)
        inflight, got = [], []
        for i in range(12):
            inflight.append((i % 4, runner.forward_deferred(examples[i % 4])))
            if len(inflight) > 2:
                k, h = inflight.pop(0)
                got.append((k, h.result()))
        got += [(k, h.result()) for k, h in inflight]
    assert len(got) == 12
    for k, out in got:
        _same(want[k], out)


def test_frame_graph_capture_beside_a_live_process_group():
    """what every rank of `bench.py --gpus N` does: an RCCL process group (its watchdog thread polling events) is
    alive while the frame graphs are captured and replayed; collectives before and after still work"""
    import os
    import socket
    import torch.distributed as dist
    from pillarnet_lts_amd import FrameGraphs, presets, synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)
        dist.barrier()
        torch.manual_seed(0)
        model = presets.build(presets.nusc_pillarnet18())
        synth.randomize_bn_(model, seed=1)
        model.to_inference(torch.bfloat16)
        frames = [torch.from_numpy(synth.nusc_frame(10, 6000, seed=s)).cuda() for s in (21, 22)]
        exs = [{"points": [f], "metadata": [{"token": i}]} for i, f in enumerate(frames)]
        graphs = FrameGraphs(model, bucket=8192, lanes=2)
        with torch.no_grad():
            want = [model(e, return_loss=False) for e in exs]
            for rep in range(3):
                handles = [graphs.forward_deferred(e) for e in exs]
                dist.all_reduce(t)                       # a collective queued between the replays
                for w, h in zip(want, handles):
                    _same(w, h.result())
        dist.barrier()
        torch.cuda.synchronize()
        assert float(t[0]) == 1.0
    finally:
        dist.destroy_process_group()


def test_two_stage_frame_graph_matches_eager():
    """Pillar R-CNN through FrameGraphs: first stage, device-side RoI padding, second stage and scores in one graph"""
    from test_rcnn_gpu import small_rcnn
    from pillarnet_lts_amd import FrameGraphs, presets, synth
    torch.manual_seed(3)
    model = presets.build(small_rcnn())
    synth.randomize_bn_(model, seed=2)
    model.to_inference(torch.bfloat16)
    import numpy as np
    scale = np.array([0.17, 0.17, 1, 1, 1], np.float32)        # the frame squeezed into the small range
    frames = [torch.from_numpy(synth.waymo_frame(9000, seed=s)[:, :5] * scale).cuda() for s in (31, 32, 33)]
    frames[1] = frames[1][:7777]
    exs = [{"points": [f], "metadata": [{"token": i}]} for i, f in enumerate(frames)]
    graphs = FrameGraphs(model, bucket=4096, lanes=2)
    with torch.no_grad():
        want = [model(e, return_loss=False) for e in exs]
        assert all(w[0]["scores"].shape[0] > 0 for w in want)
        for rep in range(2):
            handles = [graphs.forward_deferred(e) for e in exs]
            for w, h in zip(want, handles):
                _same(w, h.result())


def test_frame_graph_degenerate_frames():
    """an empty frame, a frame with every point out of range and a one-point frame replay a graph captured on an
    ordinary frame of the same bucket (row counts of 0 reach every kernel of the graph)"""
    from pillarnet_lts_amd import FrameGraphs, presets, synth
    torch.manual_seed(0)
    model = presets.build(presets.nusc_pillarnet18())
    synth.randomize_bn_(model, seed=1)
    model.to_inference(torch.bfloat16)
    graphs = FrameGraphs(model, bucket=8192, lanes=2)
    full = torch.from_numpy(synth.nusc_frame(1, 6000, seed=41)).cuda()
    frames = [full, full[:0], torch.full((50, 5), 400.0, device="cuda"),
              torch.tensor([[0.1, 0.2, -1.0, 3.0, 0.0]], device="cuda"), full]
    with torch.no_grad():
        for i, f in enumerate(frames):
            ex = {"points": [f], "metadata": [{"token": i}]}
            want = model(ex, return_loss=False)
            for rep in range(2):                      # once per lane
                _same(want, graphs(ex))
        assert sorted({k for ln in graphs.lanes for k in ln.frames}) == [(5, (8192,))]


def test_index_chain_on_a_side_stream_changes_nothing():
    """sparse_conv.RULES_AHEAD (FrameGraphs(rules_ahead=True)): the strided levels' index chain runs on a side
    stream beside level 1 -- eager and captured, replayed with eager work in between, the detections are the same bits"""
    from pillarnet_lts_amd import FrameGraphs, presets, sparse_conv, synth
    torch.manual_seed(0)
    model = presets.build(presets.nusc_pillarnet18())
    synth.randomize_bn_(model, seed=1)
    model.to_inference(torch.bfloat16)
    pts = [torch.from_numpy(synth.nusc_frame(10, 8000, seed=s)).cuda() for s in (21, 22, 23)]
    on = FrameGraphs(model, bucket=8192, lanes=1, rules_ahead=True)
    off = FrameGraphs(model, bucket=8192, lanes=1)
    assert on.rules_ahead and not off.rules_ahead
    with torch.no_grad():
        for rep in range(2):
            for i, f in enumerate(pts):
                ex = {"points": [f], "metadata": [{"token": i}]}
                want = model(ex, return_loss=False)
                assert want[0]["scores"].shape[0] > 0
                sparse_conv.RULES_AHEAD = True
                try:
                    eager_ahead = model(ex, return_loss=False)
                finally:
                    sparse_conv.RULES_AHEAD = False
                _same(want, eager_ahead)
                _same(want, on(ex))
                _same(want, off(ex))
