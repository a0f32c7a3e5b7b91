"""GPU parity of the rotated NMS / IoU kernels (through the C ABI) against the C oracle, which is
itself pinned bit-for-bit to the compiled reference (tests/test_oracle_cpu.py).

Bar: keep lists bit-exact.  IoU values: fp32 within 2e-5 abs (device cosf/sinf/atan2f differ
from glibc's in the last ulp).  A keep list can only differ if some consulted pair has
|IoU - thr| below that; a case whose oracle-reported gap is smaller must still agree everywhere such
borderline pairs do not explain the difference (tests/parity_util.py)."""
import numpy as np
import pytest
import torch

from oracle import iou3d
from pillarnet_lts_amd import iou3d_nms, synth
from parity_util import assert_keep_lists_match

pytestmark = pytest.mark.gpu
IOU_TOL = 2e-5


def _nms_case(cuda, boxes, thr):
    want, gap = iou3d.nms(boxes, thr)
    n = boxes.shape[0]
    keep = torch.LongTensor(max(n, 1))
    m = iou3d_nms.nms_gpu_raw(torch.from_numpy(boxes).to(cuda), keep, thr)
    got = keep[:m].numpy()
    if gap >= 10 * IOU_TOL:
        np.testing.assert_array_equal(got, want)
    else:
        # some consulted pair sits within the IoU tolerance of the threshold (libm last-ulp territory):
        # the lists may differ only where such pairs explain it
        assert_keep_lists_match(got, want, boxes, thr)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 128, 129, 500, 1000])
@pytest.mark.parametrize("thr", [0.2, 0.55, 0.8])
def test_nms_sizes(cuda, n, thr):
    _nms_case(cuda, synth.nms_boxes(n, seed=400 + n, cluster=1.5), thr)


def test_nms_sparse_scene(cuda):
    _nms_case(cuda, synth.nms_boxes(1000, seed=400), 0.2)


def test_nms_dense_cluster_4096(cuda):
    _nms_case(cuda, synth.nms_boxes(4096, seed=7, extent=30.0, cluster=1.0), 0.55)


@pytest.mark.parametrize("n", [4097, 5000, 9000])
def test_nms_beyond_4096_boxes(cuda, n):
    """the reference's nms_gpu takes any n (iou3d_nms.cpp:113-159); here one wave's removal words cover 16 384."""
    _nms_case(cuda, synth.nms_boxes(n, seed=11 + n, extent=60.0, cluster=1.2), 0.55)


def test_nms_size_limit_is_an_error_not_a_truncation(cuda):
    from pillarnet_lts_amd import _lib
    keep = torch.LongTensor(16385)
    with pytest.raises(_lib.PillarnetHipError, match="16384"):
        iou3d_nms.nms_gpu_raw(torch.zeros((16385, 7), device=cuda), keep, 0.2)


def test_nms_all_identical(cuda):
    b = np.tile(np.array([[1, 2, 0, 4, 2, 1.5, 0.3]], np.float32), (200, 1))
    _nms_case(cuda, b, 0.2)


def test_nms_empty(cuda):
    keep = torch.LongTensor(1)
    assert iou3d_nms.nms_gpu_raw(torch.zeros((0, 7), device=cuda), keep, 0.2) == 0


def test_iou_matrix(cuda):
    a = synth.nms_boxes(300, seed=1, cluster=2.0)
    b = synth.nms_boxes(200, seed=2, cluster=2.0)
    b[:100, :2] = a[:100, :2] + 0.3
    want = iou3d.iou_matrix(a, b)
    got = iou3d_nms.boxes_iou_bev(torch.from_numpy(a).to(cuda), torch.from_numpy(b).to(cuda)).cpu().numpy()
    assert (want > 0).mean() > 0.01
    np.testing.assert_allclose(got, want, rtol=0, atol=IOU_TOL)
    want_ov = iou3d.iou_matrix(a, b, overlap_only=True)
    got_ov = iou3d_nms.boxes_overlap_bev(torch.from_numpy(a).to(cuda), torch.from_numpy(b).to(cuda)).cpu().numpy()
    np.testing.assert_allclose(got_ov, want_ov, rtol=1e-5, atol=1e-4)


def test_cpu_named_iou_entries(cuda):
    """boxes_iou_bev_cpu / boxes_aligned_iou_bev_cpu: host tensors in and out, values of the reference's iou_bev"""
    a = synth.nms_boxes(150, seed=5, cluster=2.0)
    b = synth.nms_boxes(150, seed=6, cluster=2.0)
    b[:80, :2] = a[:80, :2] + 0.25
    want = iou3d.iou_matrix(a, b)
    ans = torch.zeros(150, 150)
    assert iou3d_nms.boxes_iou_bev_cpu(torch.from_numpy(a), torch.from_numpy(b), ans) == 1
    np.testing.assert_allclose(ans.numpy(), want, rtol=0, atol=IOU_TOL)
    ans1 = torch.zeros(150, 1)
    assert iou3d_nms.boxes_aligned_iou_bev_cpu(torch.from_numpy(a), torch.from_numpy(b), ans1) == 1
    np.testing.assert_allclose(ans1.numpy()[:, 0], np.diag(want), rtol=0, atol=IOU_TOL)
    assert (np.diag(want) > 0).sum() > 40


def test_pybind_shaped_module(cuda):
    """iou3d_nms_cuda.*: out-parameter style of iou3d_nms_api.cpp:11-19"""
    m = iou3d_nms.iou3d_nms_cuda
    a = synth.nms_boxes(90, seed=8, cluster=2.0)
    b = synth.nms_boxes(70, seed=9, cluster=2.0)
    b[:50, :2] = a[:50, :2] + 0.2
    ta, tb = torch.from_numpy(a).to(cuda), torch.from_numpy(b).to(cuda)
    ans = torch.zeros(90, 70, device=cuda)
    assert m.boxes_iou_bev_gpu(ta, tb, ans) == 1
    np.testing.assert_allclose(ans.cpu().numpy(), iou3d.iou_matrix(a, b), rtol=0, atol=IOU_TOL)
    assert m.boxes_overlap_bev_gpu(ta, tb, ans) == 1
    want_ov = iou3d.iou_matrix(a, b, overlap_only=True)
    np.testing.assert_allclose(ans.cpu().numpy(), want_ov, rtol=1e-5, atol=1e-4)
    al = torch.zeros(70, 1, device=cuda)
    assert m.boxes_aligned_overlap_bev_gpu(ta[:70], tb, al) == 1
    np.testing.assert_allclose(al.cpu().numpy()[:, 0], np.diag(want_ov[:70]), rtol=1e-5, atol=1e-4)
    keep = torch.LongTensor(90)
    n = m.nms_gpu(ta, keep, 0.2)
    assert_keep_lists_match(keep[:n].numpy(), iou3d.nms(a, 0.2)[0], a, 0.2)
    assert m.nms_normal_gpu(ta, keep, 0.3) > 0


def test_batched_sets_with_device_counts(cuda):
    S, cap = 6, 1000
    sets = [synth.nms_boxes(cap, seed=50 + s, cluster=1.5) for s in range(S)]
    counts = [1000, 0, 1, 64, 777, 130]
    thr = [0.2, 0.2, 0.55, 0.8, 0.2, 0.55]
    post = [83, 83, 83, 10, 1000, 83]
    boxes = torch.from_numpy(np.stack(sets)).to(cuda)
    keep, num = iou3d_nms.nms_rotated_sets(
        boxes, torch.tensor(counts, dtype=torch.int32, device=cuda), 0.0, None,
        torch.tensor(thr, dtype=torch.float32, device=cuda),
        torch.tensor(post, dtype=torch.int32, device=cuda))
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for s in range(S):
        want, gap = iou3d.nms(sets[s][: counts[s]], thr[s])
        want = want[: post[s]]
        if gap < 10 * IOU_TOL and not (num[s] == len(want) and np.array_equal(keep[s, : num[s]], want)):
            # compare against the untruncated oracle list up to the last position the product kept: a flipped
            # decision shifts where the post_max cut falls
            full, _ = iou3d.nms(sets[s][: counts[s]], thr[s])
            g = keep[s, : num[s]]
            cmp = full[full <= g[-1]] if (num[s] == post[s] and len(g)) else full
            assert_keep_lists_match(g, cmp, sets[s][: counts[s]], thr[s])
            continue
        assert num[s] == len(want)
        np.testing.assert_array_equal(keep[s, : num[s]], want)


def test_nms_properties_full_size(cuda):
    """Size-independent checks at pre_max 1000: kept boxes are mutually below threshold, every
    dropped box is above threshold with an earlier kept box (computed with the device IoU)."""
    b = synth.nms_boxes(1000, seed=11, cluster=1.0)
    tb = torch.from_numpy(b).to(cuda)
    keep = torch.LongTensor(1000)
    m = iou3d_nms.nms_gpu_raw(tb, keep, 0.2)
    k = keep[:m].numpy()
    assert np.all(np.diff(k) > 0)
    iou = iou3d_nms.boxes_iou_bev(tb, tb).cpu().numpy()
    sub = iou[np.ix_(k, k)]
    assert (np.triu(sub, 1) <= 0.2).all()
    dropped = np.setdiff1d(np.arange(1000), k)
    for d in dropped:
        assert (iou[k[k < d], d] > 0.2).any()


@pytest.mark.parametrize("n", [1, 65, 700, 5000])
def test_nms_normal_matches_the_reference_definition(cuda, n):
    """nms_normal_gpu (iou3d_nms.cpp:162-215): greedy NMS on the axis-aligned IoU of iou3d_nms_kernel.cu:327-338,
    restated here in float32 numpy (same operation order: the keep lists must be equal)."""
    b = synth.nms_boxes(n, seed=90 + n, extent=40.0, cluster=1.5)
    scores = np.linspace(1.0, 0.0, n).astype(np.float32)      # already sorted
    f = np.float32
    x1, x2 = b[:, 0] - b[:, 3] / f(2), b[:, 0] + b[:, 3] / f(2)
    y1, y2 = b[:, 1] - b[:, 4] / f(2), b[:, 1] + b[:, 4] / f(2)
    area = b[:, 3] * b[:, 4]
    alive, want = np.ones(n, bool), []
    for i in range(n):
        if not alive[i]:
            continue
        want.append(i)
        w = np.maximum(np.minimum(x2[i], x2) - np.maximum(x1[i], x1), f(0))
        h = np.maximum(np.minimum(y2[i], y2) - np.maximum(y1[i], y1), f(0))
        inter = w * h
        iou = inter / np.maximum(area[i] + area - inter, f(1e-8))
        alive[(iou > f(0.3)) & (np.arange(n) > i)] = False
    got, _ = iou3d_nms.nms_normal_gpu(torch.from_numpy(b).to(cuda), torch.from_numpy(scores).to(cuda), 0.3)
    np.testing.assert_array_equal(got.cpu().numpy(), np.asarray(want))
