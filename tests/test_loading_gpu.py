"""pn_merge_sweeps / loading.merge_sweeps / LoadPointCloudFromFile on the GPU against the reference's own output
(tests/golden/loading.npz) and the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import loading_oracle as lo
from pillarnet_lts_amd import loading

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _golden():
    g = np.load(os.path.join(G, "loading.npz"))
    order = [int(i) for i in g["order"]]
    raws = [g["raw.0"]] + [g[f"raw.{i + 1}"] for i in order]
    mats = [None] + [None if np.isnan(g[f"transform.{i}"]).all() else g[f"transform.{i}"] for i in order]
    lags = [0.0] + [float(g[f"time_lag.{i}"]) for i in order]
    return g, raws, mats, lags


def _close(got, want):
    """columns 3 (intensity) and 4 (time lag) and untransformed rows exact; transformed xyz to one fp32 ulp
    (float64 accumulation order of BLAS vs the kernel's left-to-right sum)."""
    assert got.shape == want.shape and got.dtype == np.float32
    np.testing.assert_array_equal(got[:, 3:], want[:, 3:])
    np.testing.assert_allclose(got[:, :3], want[:, :3], rtol=1.2e-7, atol=1e-6)


def test_merge_matches_reference_output(cuda):
    g, raws, mats, lags = _golden()
    got = loading.merge_sweeps(raws, mats, lags, cuda).cpu().numpy()
    _close(got, g["combined"])
    n_key = raws[0].shape[0]
    np.testing.assert_array_equal(got[:n_key, :4], raws[0][:, :4])      # the key sweep passes through untouched
    assert (got[:n_key, 4] == 0).all()


def test_pipeline_class_reads_bin_files_in_the_reference_order(cuda, tmp_path):
    g, _, _, _ = _golden()
    paths = []
    for i in range(4):
        p = tmp_path / f"s{i}.bin"
        g[f"raw.{i}"].tofile(p)
        paths.append(str(p))
    sweeps = [{"lidar_path": paths[i + 1],
               "transform_matrix": None if np.isnan(g[f"transform.{i}"]).all() else g[f"transform.{i}"],
               "time_lag": float(g[f"time_lag.{i}"])} for i in range(3)]
    np.random.seed(33)                                   # the seed the fixture was generated under
    res, _ = loading.LoadPointCloudFromFile(dataset="NuScenesDataset", device=cuda)(
        {"lidar": {"nsweeps": 4}}, {"lidar_path": paths[0], "sweeps": sweeps})
    assert res["lidar"]["combined"].is_cuda and res["type"] == "NuScenesDataset"
    _close(res["lidar"]["combined"].cpu().numpy(), g["combined"])
    with pytest.raises(AssertionError):
        loading.LoadPointCloudFromFile()({"lidar": {"nsweeps": 3}}, {"lidar_path": paths[0], "sweeps": sweeps})


@pytest.mark.parametrize("n_sweeps,n_pts", [(10, 30000), (1, 777), (3, 0)])
def test_merge_full_size_and_edges_vs_oracle(cuda, n_sweeps, n_pts):
    rs = np.random.RandomState(n_sweeps)
    raws, mats, lags = [], [], []
    for i in range(n_sweeps):
        n = n_pts if i != 1 else max(n_pts - 5, 0)       # ragged sweeps; (3, 0): every sweep empty
        raw = np.concatenate([rs.randn(n, 3) * [30.0, 30.0, 2.0], rs.rand(n, 2) * 100], axis=1).astype(np.float32)
        raw[: n // 20, :2] *= 0.02
        raws.append(raw)
        T = np.eye(4)
        T[:3, :3] += rs.randn(3, 3) * 0.01
        T[:3, 3] = rs.randn(3)
        mats.append(None if i == 0 else T)
        lags.append(0.05 * i)
    want = lo.combined(raws, mats, lags)
    buf, count = loading.merge_sweeps(raws, mats, lags, cuda, exact_rows=False)
    n = int(count.item())
    assert n == want.shape[0]
    if n:
        _close(buf[:n].cpu().numpy(), want)
    # order property at full size: time lags never decrease along the rows (sweeps stay in order)
    assert (np.diff(buf[:n, 4].cpu().numpy()) >= 0).all()


def test_merge_rejects_host_output():
    from pillarnet_lts_amd._lib import PillarnetHipError
    with pytest.raises(PillarnetHipError):
        loading.merge_sweeps([np.zeros((3, 5), np.float32)], [None], [0.0], "cpu")


# ---- Waymo branch (loading.py:63-100,142-175) ------------------------------------------------------------------
def _waymo_golden():
    g = np.load(os.path.join(G, "loading_waymo.npz"))
    frames = [(g[f"xyz.{i}"], g[f"feat.{i}"]) for i in range(3)]
    mats = [None, g["transform.1"], None]                 # the third frame carries no transform (nan block in the fixture)
    assert np.isnan(g["transform.2"]).all()
    lags = [0.0, float(g["time_lag.1"]), float(g["time_lag.2"])]
    return g, frames, mats, lags


def _close_waymo(got, want):
    """elongation / time lag / untransformed xyz exact; transformed xyz to one fp32 ulp (as the nuScenes branch);
    tanh(intensity): device tanhf against numpy's, 2 ulp of a value below 1 (2.4e-7)."""
    assert got.shape == want.shape and got.dtype == np.float32
    np.testing.assert_array_equal(got[:, 4:], want[:, 4:])
    np.testing.assert_allclose(got[:, :3], want[:, :3], rtol=1.2e-7, atol=1e-6)
    np.testing.assert_allclose(got[:, 3], want[:, 3], rtol=0, atol=2.4e-7)


def test_waymo_pipeline_matches_reference_output(cuda, tmp_path):
    """LoadPointCloudFromFile(dataset="WaymoDataset") on frame pickles this test writes (the converter's layout) against
    what the reference's own class returned for the same arrays (tests/golden/loading_waymo.npz): one sweep -> `points`
    only; three sweeps in the listed order -> `combined`, nothing filtered, intensity through tanh."""
    import pickle
    g, frames, mats, lags = _waymo_golden()
    paths = []
    for i, (xyz, feat) in enumerate(frames):
        p = tmp_path / f"frame{i}.pkl"
        with open(p, "wb") as f:
            pickle.dump({"lidars": {"points_xyz": xyz.copy(), "points_feature": feat.copy()}}, f)
        paths.append(str(p))
    step = loading.LoadPointCloudFromFile(dataset="WaymoDataset", device=cuda)
    one, _ = step({"lidar": {"nsweeps": 1}}, {"path": paths[0]})
    assert one["type"] == "WaymoDataset" and "combined" not in one["lidar"] and one["lidar"]["points"].is_cuda
    _close_waymo(one["lidar"]["points"].cpu().numpy(), g["one.points"])
    sweeps = [{"path": paths[i], "transform_matrix": mats[i], "time_lag": lags[i]} for i in (1, 2)]
    three, _ = step({"lidar": {"nsweeps": 3}}, {"path": paths[0], "sweeps": sweeps})
    _close_waymo(three["lidar"]["combined"].cpu().numpy(), g["three.combined"])
    np.testing.assert_array_equal(three["lidar"]["times"].cpu().numpy(), g["three.times"])
    assert three["lidar"]["points"].shape == g["three.points"].shape
    n0 = frames[0][0].shape[0]
    assert three["lidar"]["combined"].shape[0] == sum(f[0].shape[0] for f in frames)          # no close-point filter
    np.testing.assert_array_equal(three["lidar"]["combined"][:n0, :3].cpu().numpy(), frames[0][0])
    with pytest.raises(AssertionError):
        step({"lidar": {"nsweeps": 2}}, {"path": paths[0], "sweeps": sweeps})
    with pytest.raises(NotImplementedError):
        loading.LoadPointCloudFromFile(dataset="KittiDataset", device=cuda)({"lidar": {"nsweeps": 1}}, {})


def test_waymo_merge_full_size_vs_oracle(cuda):
    """configs[2]-sized frame (180 000 points) + two earlier sweeps against the numpy oracle"""
    rs = np.random.RandomState(4)
    frames, mats, lags = [], [], []
    for i in range(3):
        n = 180000 - 777 * i
        frames.append(((rs.randn(n, 3) * [40.0, 40.0, 2.0]).astype(np.float32),
                       np.stack([np.exp(rs.randn(n) * 2), rs.rand(n)], 1).astype(np.float32)))
        T = np.eye(4)
        T[:3, :3] += rs.randn(3, 3) * 0.01
        T[:3, 3] = rs.randn(3)
        mats.append(None if i == 0 else T)
        lags.append(0.1 * i)
    want = lo.combined_waymo(frames, mats, lags)
    raws = [np.concatenate(f, axis=1) for f in frames]
    got = loading.merge_sweeps(raws, mats, lags, cuda, num_point_feature=5, filter_close=[0, 0, 0], tanh_column=3)
    _close_waymo(got.cpu().numpy(), want)
