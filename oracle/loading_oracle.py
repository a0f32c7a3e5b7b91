"""CPU restatement of the reference's multi-sweep loading (numpy).  TEST INFRASTRUCTURE ONLY.

Follows det3d/datasets/pipelines/loading.py:38-61 (remove_close, read_sweep) and :118-140 (the nuScenes
branch of LoadPointCloudFromFile.__call__), and :63-94 (read_single_waymo, read_single_waymo_sweep) with :142-175
(the Waymo branch), on in-memory sweeps instead of file paths.  Pinned by tests/golden/loading.npz and
tests/golden/loading_waymo.npz, produced by the reference's own functions on files written from the same
arrays (tests/golden/make_golden.py::golden_loading, golden_loading_waymo)."""
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


def waymo_frame(xyz, feat, transform=None):
    """read_single_waymo / read_single_waymo_sweep: (xyz, tanh(intensity), elongation) [n, 5], moved by the 4x4."""
    feat = np.array(feat, np.float32)
    feat[:, 0] = np.tanh(feat[:, 0])
    pts = np.concatenate([np.asarray(xyz, np.float32), feat], axis=-1).T
    if transform is not None:
        pts[:3, :] = np.asarray(transform).dot(np.vstack((pts[:3, :], np.ones(pts.shape[1]))))[:3, :]
    return pts.T


def combined_waymo(frames, transforms, time_lags):
    """frames: list of (xyz [n, 3], feat [n, 2]), key frame first.  One frame -> points [n, 5]; several -> the
    `combined` cloud [N, 6] (points + time lag), sweeps in the listed order, nothing filtered."""
    pts = [waymo_frame(*frames[0])]
    if len(frames) == 1:
        return pts[0]
    tms = [np.zeros((pts[0].shape[0], 1))]
    for (xyz, feat), T, lag in zip(frames[1:], transforms[1:], time_lags[1:]):
        pts.append(waymo_frame(xyz, feat, T))
        tms.append(lag * np.ones((pts[-1].shape[0], 1)))
    points = np.concatenate(pts, axis=0)
    return np.hstack([points, np.concatenate(tms, axis=0).astype(points.dtype)])
