"""GPU parity of the fused reader (pn_pillarize through the C ABI) against the CPU oracle.
Indices / point->pillar map: bit-exact.  Features: fp32, |err| <= 1e-5 + 1e-5*|ref| (the 7-term
dot product is summed in a different order than the reference's GEMM)."""
import numpy as np
import pytest
import torch

from oracle import pillar_oracle as po
from pillarnet_lts_amd import pillar_ops, synth

pytestmark = pytest.mark.gpu

NUSC = dict(pillar_size=0.075, pc_range=[-54, -54, -5.0, 54, 54, 3.0])
WAYMO = dict(pillar_size=0.1, pc_range=[-75.2, -75.2, -2, 75.2, 75.2, 4])
TINY = dict(pillar_size=0.5, pc_range=[-8, -8, -3, 8, 8, 1])


def _weights(F, C=32, seed=0):
    rng = np.random.default_rng(seed)
    w = rng.normal(0, (2.0 / (F + 2)) ** 0.5, (C, F + 2)).astype(np.float32)
    gamma = rng.uniform(0.8, 1.2, C).astype(np.float32)
    beta = rng.normal(0, 0.1, C).astype(np.float32)
    mean = rng.normal(0, 0.1, C).astype(np.float32)
    var = rng.uniform(0.5, 1.5, C).astype(np.float32)
    return w, gamma, beta, mean, var


def _run(points_list, grid, cuda, true_divide=False, seed=0):
    F = points_list[0].shape[1]
    w, gamma, beta, mean, var = _weights(F, seed=seed)
    ref = po.pillarize(points_list, grid["pc_range"], grid["pillar_size"], w, gamma, beta, mean,
                       var, 1e-3, true_divide)
    scale, shift = po.fold_bn(gamma, beta, mean, var, 1e-3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    sp, ppi = pillar_ops.pillarize([t(p) for p in points_list], t(w), t(scale), t(shift),
                                   grid["pillar_size"], grid["pc_range"], want_point_index=True,
                                   true_divide=true_divide)
    torch.cuda.synchronize()
    M = sp.num_active_host()
    assert M == ref["indices"].shape[0]
    np.testing.assert_array_equal(sp.indices.cpu().numpy(), ref["indices"])
    np.testing.assert_array_equal(ppi.cpu().numpy(), ref["point_pillar_index"])
    got = sp.features.cpu().numpy()
    assert got.shape == ref["features"].shape
    np.testing.assert_allclose(got, ref["features"], rtol=1e-5, atol=1e-5)
    # structural properties that hold at any size
    if M:
        idx = ref["indices"].astype(np.int64)
        H, W = po.bev_spatial_shape(grid["pillar_size"], grid["pc_range"])
        flat = idx[:, 0] * H * W + idx[:, 1] * W + idx[:, 2]
        assert np.all(np.diff(flat) > 0)
        assert (got >= 0).all()
    return sp, ref


def test_tiny_grid_two_samples(cuda):
    rng = np.random.default_rng(1)
    a = rng.uniform(-9, 9, (2000, 5)).astype(np.float32)
    b = rng.uniform(-9, 9, (777, 5)).astype(np.float32)
    _run([a, b], TINY, cuda)


def test_nusc_single_sweep(cuda):
    _run([synth.nusc_sweep(30000, seed=0)], NUSC, cuda)


def test_nusc_10_sweeps_full_size(cuda):
    sp, ref = _run([synth.nusc_frame(10, 30000, seed=0)], NUSC, cuda)
    assert ref["indices"].shape[0] > 50000


def test_nusc_batch4(cuda):
    _run([synth.nusc_sweep(20000 + 1000 * k, seed=10 + k) for k in range(4)], NUSC, cuda)


def test_waymo_grid(cuda):
    _run([synth.waymo_frame(180000, seed=100)], WAYMO, cuda)


def test_true_divide_mode(cuda):
    _run([synth.nusc_sweep(30000, seed=3)], NUSC, cuda, true_divide=True)


def test_empty_and_out_of_range(cuda):
    far = np.full((100, 5), 500.0, dtype=np.float32)
    sp, _ = _run([far], TINY, cuda)
    assert sp.num_active_host() == 0
    empty = np.zeros((0, 5), dtype=np.float32)
    rng = np.random.default_rng(2)
    some = rng.uniform(-8, 8, (50, 5)).astype(np.float32)
    _run([empty, some, empty], TINY, cuda)


def test_single_point_and_duplicates(cuda):
    one = np.array([[0.1, 0.2, 0.3, 4.0, 0.0]], dtype=np.float32)
    _run([one], TINY, cuda)
    dup = np.repeat(one, 300, axis=0)
    dup[:, 2] = np.linspace(-3, 1, 300)
    _run([dup], TINY, cuda)


def test_cell_boundaries(cuda):
    # points exactly on / one ulp either side of every cell edge of the tiny grid
    edges = np.arange(-8, 8.5, 0.5, dtype=np.float32)
    xs = np.concatenate([edges, np.nextafter(edges, np.float32(100)),
                         np.nextafter(edges, np.float32(-100))])
    pts = np.zeros((xs.size * 2, 5), dtype=np.float32)
    pts[: xs.size, 0] = xs
    pts[xs.size:, 1] = xs
    _run([pts], TINY, cuda)
    _run([pts], TINY, cuda, true_divide=True)
    # nuScenes grid: 0.075 is not representable, so this is where *inv and /ps can differ
    e = (np.arange(0, 1441, dtype=np.float64) * 0.075 - 54.0).astype(np.float32)
    xs = np.concatenate([e, np.nextafter(e, np.float32(100)), np.nextafter(e, np.float32(-100))])
    pts = np.zeros((xs.size, 5), dtype=np.float32)
    pts[:, 0] = xs
    pts[:, 1] = xs[::-1]
    _run([pts], NUSC, cuda)
    _run([pts], NUSC, cuda, true_divide=True)


def test_matches_torch_device_expression(cuda):
    """The cell arithmetic must be what torch itself computes on this GPU for the expression in
    dynamic_pillar_encoder.py:34 (plain PyTorch, not reference code)."""
    pts = synth.nusc_frame(10, 30000, seed=5)
    x = torch.from_numpy(pts[:, 0]).to(cuda)
    got = ((x - NUSC["pc_range"][0]) / NUSC["pillar_size"]).floor().int().cpu().numpy()
    want = po.cell_coords(pts[:, 0], NUSC["pc_range"][0], NUSC["pillar_size"], true_divide=False)
    np.testing.assert_array_equal(got, want)


def test_inf_and_huge_coordinates(cuda):
    # (NaN coordinates are left unspecified: the reference would place them in cell 0 and
    # propagate NaN features through its CAS max in an order-dependent way)
    pts = synth.nusc_sweep(1000, seed=7)
    pts[6, 1] = np.inf
    pts[7, 0] = -np.inf
    pts[8, 0] = 3e38
    _run([pts], NUSC, cuda)
