"""Host side of the rotated IoU / NMS operators (mirror of det3d/ops/iou3d_nms/iou3d_nms_utils.py
and of the `iou3d_nms_cuda` pybind surface, iou3d_nms_api.cpp:11-19), on the HIP C ABI."""
import numpy as np
import torch

from . import _lib


def _need_cuda(t, what):
    if t.device.type != "cuda":
        raise _lib.PillarnetHipError(f"{what}: tensor must live on the GPU (no CPU fallback)")


def nms_rotated_sets(boxes, counts=None, thresh=0.2, post_max=None, thresh_per_set=None,
                     post_max_per_set=None):
    """Batched device-side NMS.  boxes [S, cap, 7] fp32 (pcdet frame, sorted by score), counts
    int32[S] on the device (None: all cap boxes valid).  Returns (keep int32 [S, stride],
    num_keep int32 [S]) on the device; nothing is synchronised."""
    _need_cuda(boxes, "nms_rotated_sets")
    assert boxes.dim() == 3 and boxes.shape[2] == 7 and boxes.dtype == torch.float32
    boxes = boxes.contiguous()
    S, cap = boxes.shape[0], boxes.shape[1]
    dev = boxes.device
    stride = cap if post_max is None else max(1, min(cap, int(post_max)))
    if post_max_per_set is not None:
        stride = cap
    keep = torch.empty((S, stride), dtype=torch.int32, device=dev)
    num = torch.zeros(S, dtype=torch.int32, device=dev)
    if cap == 0:
        return keep, num
    L = _lib.lib()
    wsb = L.pn_nms_workspace_bytes(S, cap)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    rc = L.pn_nms_rotated(_lib.ptr(boxes), S, cap, _lib.ptr(counts), cap if counts is None else 0,
                          _lib.ptr(thresh_per_set), float(thresh), _lib.ptr(post_max_per_set),
                          stride, stride, _lib.ptr(keep), _lib.ptr(num), _lib.ptr(ws), wsb,
                          _lib.current_stream(dev))
    _lib.check(rc, "pn_nms_rotated")
    return keep, num


def nms_gpu_raw(boxes, keep, thresh):
    """`iou3d_nms_cuda.nms_gpu(boxes, keep, thresh) -> num_kept` (iou3d_nms.cpp:113-159):
    boxes [n,7] on the device sorted by score, keep a *CPU* int64 tensor that receives the kept
    positions.  This reference-shaped entry synchronises, as the reference's does."""
    n = boxes.shape[0]
    if n == 0:
        return 0
    k, num = nms_rotated_sets(boxes.view(1, n, 7).float(), None, thresh, None)
    m = int(num.item())
    keep[:m] = k[0, :m].to("cpu", torch.int64)
    return m


def to_pcdet(boxes):
    """iou3d_nms_utils.py:30-34: det3d box (.., dx<->dy swapped, heading -> -heading - pi/2)."""
    boxes = boxes[:, [0, 1, 2, 4, 3, 5, -1]]
    boxes[:, -1] = -boxes[:, -1] - np.pi / 2
    return boxes


def _pairwise(a, b, aligned, want_iou):
    _need_cuda(a, "boxes pairwise")
    a = a.contiguous().float()
    b = b.contiguous().float()
    na, nb = a.shape[0], b.shape[0]
    out = torch.zeros((na, 1) if aligned else (na, nb), dtype=torch.float32, device=a.device)
    rc = _lib.lib().pn_boxes_pairwise_bev(_lib.ptr(a), na, _lib.ptr(b), nb, int(aligned),
                                          int(want_iou), _lib.ptr(out), _lib.current_stream(a.device))
    _lib.check(rc, "pn_boxes_pairwise_bev")
    return out


def boxes_iou_bev(boxes_a, boxes_b):
    """iou3d_nms_utils.py:13-27 -> boxes_iou_bev_gpu."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return _pairwise(boxes_a, boxes_b, False, True)


def _staged_iou(boxes_a, boxes_b, ans_iou, aligned):
    if not torch.cuda.is_available():
        raise _lib.PillarnetHipError("boxes_*_iou_bev_cpu: the rotated IoU runs on the GPU (no CPU fallback)")
    assert boxes_a.is_contiguous() and boxes_b.is_contiguous()          # CHECK_CONTIGUOUS, iou3d_cpu.cpp:237-238
    dev = torch.device("cuda", torch.cuda.current_device())
    out = _pairwise(boxes_a.to(dev), boxes_b.to(dev), aligned, True)
    ans_iou.copy_(out.view(ans_iou.shape))
    return 1


def boxes_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    """`iou3d_nms_cuda.boxes_iou_bev_cpu(a [N,7], b [M,7], ans [N,M]) -> 1` (iou3d_cpu.cpp:232-252): host tensors
    in and out.  The reference loops `iou_bev` over the pairs on the CPU; here the pairs go through the same HIP
    kernel as boxes_iou_bev_gpu on a staged copy (this package computes nothing on the CPU)."""
    return _staged_iou(boxes_a, boxes_b, ans_iou, False)


def boxes_aligned_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    """`iou3d_nms_cuda.boxes_aligned_iou_bev_cpu(a [N,7], b [N,7], ans [N,1]) -> 1` (iou3d_cpu.cpp:254-273)."""
    assert boxes_a.shape[0] == boxes_b.shape[0]
    return _staged_iou(boxes_a, boxes_b, ans_iou, True)


def boxes_overlap_bev(boxes_a, boxes_b):
    return _pairwise(boxes_a, boxes_b, False, False)


def _iou3d_from_bev(boxes_a, boxes_b, overlaps_bev, aligned):
    # iou3d_nms_utils.py:53-74 / 95-116
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    shape = (-1, 1) if aligned else (1, -1)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(*shape)
    b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(*shape)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(*shape)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:37-74 (det3d-frame boxes in, [N, M] 3-D IoU out)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    boxes_a = to_pcdet(boxes_a)
    boxes_b = to_pcdet(boxes_b)
    return _iou3d_from_bev(boxes_a, boxes_b, _pairwise(boxes_a, boxes_b, False, False), False)


def boxes_aligned_iou3d_gpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:77-116 ([N, 1])."""
    assert boxes_a.shape[0] == boxes_b.shape[0]
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    boxes_a = to_pcdet(boxes_a)
    boxes_b = to_pcdet(boxes_b)
    return _iou3d_from_bev(boxes_a, boxes_b, _pairwise(boxes_a, boxes_b, True, False), True)


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """iou3d_nms_utils.py:119-133."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    boxes = boxes[order].contiguous()
    keep = torch.LongTensor(boxes.size(0))
    num_out = nms_gpu_raw(boxes, keep, thresh)
    return order[keep[:num_out].to(boxes.device)].contiguous(), None


def nms_normal_gpu_raw(boxes, keep, thresh):
    """`iou3d_nms_cuda.nms_normal_gpu(boxes, keep, thresh) -> num_kept` (iou3d_nms.cpp:162-215)."""
    _need_cuda(boxes, "nms_normal_gpu")
    n = boxes.shape[0]
    if n == 0:
        return 0
    boxes = boxes.contiguous().float()
    L = _lib.lib()
    dev = boxes.device
    out = torch.empty(n, dtype=torch.int32, device=dev)
    num = torch.empty(1, dtype=torch.int32, device=dev)
    ws = torch.empty(int(L.pn_nms_workspace_bytes(1, n)), dtype=torch.uint8, device=dev)
    rc = L.pn_nms_normal(_lib.ptr(boxes), n, float(thresh), n, _lib.ptr(out), _lib.ptr(num), _lib.ptr(ws), ws.numel(),
                         _lib.current_stream(dev))
    _lib.check(rc, "pn_nms_normal")
    m = int(num.item())
    keep[:m] = out[:m].to("cpu", torch.int64)
    return m


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """iou3d_nms_utils.py:136-151."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    boxes = boxes[order].contiguous()
    keep = torch.LongTensor(boxes.size(0))
    num_out = nms_normal_gpu_raw(boxes, keep, thresh)
    return order[keep[:num_out].to(boxes.device)].contiguous(), None


class iou3d_nms_cuda:
    """Drop-in for the reference's `iou3d_nms_cuda` pybind module (iou3d_nms_api.cpp:11-19): same names, argument
    order and out-parameter style; boxes [n, 7] fp32 in the pcdet frame, results written into the caller's tensor."""

    @staticmethod
    def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
        ans_overlap.copy_(_pairwise(boxes_a, boxes_b, False, False))
        return 1

    @staticmethod
    def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
        ans_iou.copy_(_pairwise(boxes_a, boxes_b, False, True))
        return 1

    @staticmethod
    def boxes_aligned_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
        assert boxes_a.shape[0] == boxes_b.shape[0]
        ans_overlap.copy_(_pairwise(boxes_a, boxes_b, True, False).view(ans_overlap.shape))
        return 1

    nms_gpu = staticmethod(nms_gpu_raw)
    nms_normal_gpu = staticmethod(nms_normal_gpu_raw)
    boxes_iou_bev_cpu = staticmethod(boxes_iou_bev_cpu)
    boxes_aligned_iou_bev_cpu = staticmethod(boxes_aligned_iou_bev_cpu)
