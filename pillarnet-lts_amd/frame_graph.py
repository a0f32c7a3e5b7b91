"""One HIP graph per frame shape for PillarNet (and two-stage Pillar R-CNN) inference.

A nuScenes frame is ~75 kernel launches of 2..50 us each; queued from Python one by one they cost the host about as
long as they cost the GPU.  Every launch of the inference path is sized by capacities known on the host (point
count, grid size) and reads the live counts from device memory, so a frame with the same padded point counts is
the same launch sequence with the same arguments: it is captured once and replayed.

Points are copied into the graph's input buffer and padded to a multiple of `bucket` with points far outside the
range, which the reader's first pass drops exactly as it drops a real out-of-range point
(det3d/models/readers/dynamic_pillar_encoder.py:30-41 masks them the same way).
"""
from collections import OrderedDict

import torch

from . import _lib
from .bbox_heads import PendingDetections, ReadyDetections

FILLER = -1.0e6      # metres; outside every configured point-cloud range


class _Frame:
    __slots__ = ("graph", "points", "valid", "outs")


class _Lane:
    """a stream and the graphs captured for it (a graph's buffers belong to one frame in flight)"""
    __slots__ = ("stream", "frames")


class FrameGraphs:
    """graphs = FrameGraphs(model); graphs(example) or graphs.forward_deferred(example).result()
    return what model(example, return_loss=False) returns."""

    def __init__(self, model, bucket=16384, max_graphs=4, lanes=1, shared_cu_tiles=None, rules_ahead=None):
        """lanes > 1: consecutive frames go round robin to that many streams (default priority), each with its
        own graphs, so the thin tail of one frame (ranking, NMS: a handful of workgroups) and the latency-bound
        narrow levels share the chip with another frame's kernels.  Frames still finish in order per lane; a
        caller that keeps `lanes` frames in flight (forward_deferred) gets the overlap.
        rules_ahead (default off): the graph forks -- the index chain of the strided backbone levels runs on a side
        stream beside level 1's convolutions (sparse_conv.rules_ahead); same results, 1.3 % less device time per
        one-lane frame, but the runtime takes 1.2 ms instead of 0.2 ms of host time to launch a graph with a fork."""
        first = getattr(model, "single_det", model)          # two-stage detectors wrap the one-stage model
        if getattr(first, "compute_dtype", None) != torch.bfloat16 or model.training:
            raise ValueError("FrameGraphs needs a model prepared with to_inference(torch.bfloat16)")
        self.two_stage = first is not model
        self.model, self.bucket, self.max_graphs = model, int(bucket), int(max_graphs)
        self.lanes = []
        for _ in range(int(lanes)):
            ln = _Lane()
            ln.stream = torch.cuda.Stream() if lanes > 1 else None      # None: the caller's current stream
            ln.frames = OrderedDict()     # (feature count, padded sizes) -> _Frame, least recently used first
            self.lanes.append(ln)
        self.turn = 0
        self.shared_cu_tiles = lanes > 1 if shared_cu_tiles is None else bool(shared_cu_tiles)
        self.rules_ahead = bool(rules_ahead)

    @property
    def frames(self):
        return self.lanes[0].frames

    # the frame's launches, no host sync anywhere; None when the ranked head does not apply
    def _launch(self, points):
        # with several lanes the graphs are recorded with tiles that let two frames' workgroups share a CU
        from . import sparse_conv
        old = _lib.lib().pn_gg_set_tile_policy(1 if self.shared_cu_tiles else 0)
        old_ahead, sparse_conv.RULES_AHEAD = sparse_conv.RULES_AHEAD, self.rules_ahead
        try:
            return self._launch_frame(points)
        finally:
            _lib.lib().pn_gg_set_tile_policy(old)
            sparse_conv.RULES_AHEAD = old_ahead

    def _launch_frame(self, points):
        m = self.model
        if self.two_stage:
            return m.forward_raw(points)
        bev, _ = m.extract_feat(dict(points=points, batch_size=len(points)))
        plan = m.bbox_head.lazy_plan(m.test_cfg)
        if plan is None:
            return None
        return m.bbox_head.predict_lazy(None, bev, m.test_cfg, plan, raw=True)

    def _capture(self, frames, key, pts):
        F, sizes = key
        dev = pts[0].device
        fr = _Frame()
        fr.points = [torch.full((n, F), FILLER, dtype=torch.float32, device=dev) for n in sizes]
        fr.valid = [0] * len(sizes)
        self._fill(fr, pts)
        # eager once: kernel attributes, folded weights and launch plans are made outside the capture
        if self._launch(fr.points) is None:
            return None
        torch.cuda.synchronize(dev)
        fr.graph = torch.cuda.CUDAGraph()
        # (thread-local error mode: another thread of the process -- a collective library's watchdog polling its
        # events -- must not invalidate the capture)
        with torch.cuda.graph(fr.graph, capture_error_mode="thread_local"):
            fr.outs = self._launch(fr.points)
        while len(frames) >= self.max_graphs:
            frames.popitem(last=False)
        frames[key] = fr
        return fr

    @staticmethod
    def _fill(fr, pts):
        for b, p in enumerate(pts):
            n = p.shape[0]
            fr.points[b][:n].copy_(p)
            if n < fr.valid[b]:
                fr.points[b][n:fr.valid[b]].fill_(FILLER)
            fr.valid[b] = n

    @torch.no_grad()
    def forward_deferred(self, example):
        pts = list(example["points"])
        m = self.model
        if len(pts) == 0 or pts[0].device.type != "cuda":
            raise ValueError("FrameGraphs: points must be CUDA tensors, one per sample")
        key = (pts[0].shape[1], tuple(max(self.bucket, -(-p.shape[0] // self.bucket) * self.bucket) for p in pts))
        lane = self.lanes[self.turn]
        self.turn = (self.turn + 1) % len(self.lanes)
        if lane.stream is None:
            return self._run(lane, key, pts, example)
        caller = torch.cuda.current_stream(pts[0].device)
        lane.stream.wait_stream(caller)               # the points may have been produced on the caller's stream
        with torch.cuda.stream(lane.stream):
            pending = self._run(lane, key, pts, example)
        for t in (pending.boxes, pending.scores, pending.labels) if hasattr(pending, "boxes") else ():
            t.record_stream(caller)                   # results are consumed on the caller's stream
        return pending

    def _run(self, lane, key, pts, example):
        m = self.model
        fr = lane.frames.get(key)
        if fr is None:
            fr = self._capture(lane.frames, key, pts)
            if fr is None:                # configuration outside the ranked head: plain eager call
                return ReadyDetections(m(example, return_loss=False))
        else:
            lane.frames.move_to_end(key)
            self._fill(fr, pts)
        fr.graph.replay()
        outs = tuple(t.clone() for t in fr.outs)     # the next replay overwrites the graph's own buffers
        if self.two_stage:
            from .second_stage import PendingRefined
            return PendingRefined(example, outs, lambda: m(example, return_loss=False))
        return PendingDetections(m.bbox_head, example, outs, lambda: m(example, return_loss=False))

    def __call__(self, example, return_loss=False):
        assert not return_loss
        return self.forward_deferred(example).result()
