"""CPU restatement of the reference's multi-sweep loading (numpy).  TEST INFRASTRUCTURE ONLY.

Follows det3d/datasets/pipelines/loading.py:38-61 (remove_close, read_sweep) and :118-140 (the nuScenes
branch of LoadPointCloudFromFile.__call__), on in-memory sweeps instead of .bin paths.  Pinned by
tests/golden/loading.npz, produced by the reference's own functions on .bin files written from the same
arrays (tests/golden/make_golden.py::golden_loading)."""
import numpy as np


def remove_close(points_T, radius):
    x_filt = np.abs(points_T[0, :]) < radius
    y_filt = np.abs(points_T[1, :]) < radius
    return points_T[:, np.logical_not(np.logical_and(x_filt, y_filt))]


def read_sweep(raw, transform, time_lag, num_point_feature=4, min_distance=1.0):
    pts = remove_close(raw[:, :num_point_feature].T.copy(), min_distance)
    n = pts.shape[1]
    if transform is not None:
        pts[:3, :] = np.asarray(transform).dot(np.vstack((pts[:3, :], np.ones(n))))[:3, :]
    return pts.T, (time_lag * np.ones((1, n))).T


def combined(raw_sweeps, transforms, time_lags, num_point_feature=4):
    """key sweep first (unfiltered, dt = 0); -> float32 [N, num_point_feature + 1]."""
    key = raw_sweeps[0][:, :num_point_feature]
    pts, tms = [key], [np.zeros((key.shape[0], 1))]
    for raw, T, lag in zip(raw_sweeps[1:], transforms[1:], time_lags[1:]):
        p, t = read_sweep(raw, T, lag, num_point_feature)
        pts.append(p)
        tms.append(t)
    points = np.concatenate(pts, axis=0)
    times = np.concatenate(tms, axis=0).astype(points.dtype)
    return np.hstack([points, times])
