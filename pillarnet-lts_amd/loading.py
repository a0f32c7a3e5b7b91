"""The data step in front of the reader: multi-sweep loading of a nuScenes frame --
det3d/datasets/pipelines/loading.py:15-61 (read_file / remove_close / read_sweep) and :103-140
(LoadPointCloudFromFile, NuScenesDataset branch) -- and of a Waymo frame -- :63-100 (read_single_waymo,
read_single_waymo_sweep, get_obj) and :142-175 (WaymoDataset branch) -- with the merge done on the GPU.

The host part stays what it is in the reference (np.fromfile of the float32[n, 5] .bin sweeps, the
`np.random.choice` order of the non-key sweeps); the per-point work (close-point filter, 4x4 transform in
float64, time-lag column, concatenation) is one `pn_merge_sweeps` call on the uploaded bytes and leaves the
`[N, 5]` cloud in HBM, which is where `DynamicPFE` wants it."""
import ctypes

import numpy as np
import torch

from . import _lib
from .registry import PIPELINES


def read_file(path, num_point_feature=4):
    """loading.py:15-36 without the virtual-point branch: the raw [n, 5] rows (all five columns are kept on
    the way to the GPU; `num_point_feature` of them are used)."""
    return np.fromfile(path, dtype=np.float32).reshape(-1, 5)


def merge_sweeps(raw_sweeps, transforms, time_lags, device, num_point_feature=4, min_distance=1.0,
                 exact_rows=True, filter_close=None, tanh_column=-1):
    """raw_sweeps: list of float32 [n_i, >=num_point_feature] arrays / tensors, key sweep first.
    transforms[i]: None or 4x4 (float64) for sweep i (None for the key sweep); time_lags[i]: float.
    The key sweep is neither filtered nor moved (loading.py:118-121).  Returns the `combined` cloud
    [N, num_point_feature + 1] fp32 on `device` (loading.py:136-140); with exact_rows=False the full-capacity
    buffer and the device row count instead (no host sync).  filter_close[i] (default: every sweep but the key one)
    switches the close-point filter per sweep; tanh_column stores that feature column as tanh(value)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.PillarnetHipError("merge_sweeps: the output lives on the GPU (no CPU fallback)")
    S = len(raw_sweeps)
    assert S >= 1 and len(transforms) == S and len(time_lags) == S
    parts = [torch.as_tensor(np.ascontiguousarray(r) if isinstance(r, np.ndarray) else r, dtype=torch.float32)
             for r in raw_sweeps]
    raw_dim = parts[0].shape[1]
    assert all(p.dim() == 2 and p.shape[1] == raw_dim for p in parts) and raw_dim >= num_point_feature
    starts = np.zeros(S + 1, np.int32)
    starts[1:] = np.cumsum([p.shape[0] for p in parts])
    total = int(starts[-1])
    raw = torch.cat([p.to(dev, non_blocking=True) for p in parts]) if total else torch.empty((0, raw_dim), device=dev)
    T = np.tile(np.eye(4), (S, 1, 1))
    has_t = np.zeros(S, np.uint8)
    for i, m in enumerate(transforms):
        if m is not None:
            T[i], has_t[i] = np.asarray(m, np.float64), 1
    lag = np.asarray(time_lags, np.float64).astype(np.float32)      # `.astype(points.dtype)`, loading.py:134
    if filter_close is None:
        filt = np.ones(S, np.uint8)
        filt[0] = 0
    else:
        filt = np.asarray(filter_close, np.uint8)
        assert filt.shape == (S,)
    T_d = torch.from_numpy(T.reshape(S, 16)).to(dev)
    has_d, lag_d, filt_d = (torch.from_numpy(x).to(dev) for x in (has_t, lag, filt))
    out = torch.empty((max(total, 1), num_point_feature + 1), dtype=torch.float32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    L = _lib.lib()
    ws = torch.empty(L.pn_merge_sweeps_workspace_bytes(total), dtype=torch.uint8, device=dev)
    rc = L.pn_merge_sweeps(_lib.ptr(raw), raw_dim, num_point_feature, starts.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                           S, _lib.ptr(T_d), _lib.ptr(has_d), _lib.ptr(lag_d), _lib.ptr(filt_d), float(min_distance),
                           int(tanh_column), _lib.ptr(out), _lib.ptr(count), _lib.ptr(ws), ws.numel(), _lib.current_stream(dev))
    _lib.check(rc, "pn_merge_sweeps")
    if not exact_rows:
        return out, count
    return out[: int(count.item())]


def get_obj(path):
    """loading.py:97-100: a Waymo frame as the reference's converter stores it -- a pickled dict with
    obj["lidars"]["points_xyz"] [n, 3] and ["points_feature"] [n, 2] (intensity, elongation)."""
    import pickle
    with open(path, "rb") as f:
        return pickle.load(f)


def _waymo_raw(obj):
    lid = obj["lidars"]
    return np.concatenate([np.asarray(lid["points_xyz"], np.float32), np.asarray(lid["points_feature"], np.float32)], axis=-1)


@PIPELINES.register_module
class LoadPointCloudFromFile:
    """loading.py:103-175, NuScenesDataset and WaymoDataset branches; `res["lidar"]["points"]` (and, for several
    sweeps, `combined` / `times`) are CUDA tensors."""

    def __init__(self, dataset="NuScenesDataset", device="cuda", **kwargs):
        self.type = dataset
        self.device = device
        self.random_select = kwargs.get("random_select", False)
        self.npoints = kwargs.get("npoints", 16834)

    def __call__(self, res, info):
        res["type"] = self.type
        if self.type == "NuScenesDataset":
            return self._nuscenes(res, info)
        if self.type == "WaymoDataset":
            return self._waymo(res, info)
        raise NotImplementedError

    def _nuscenes(self, res, info):
        if res.get("virtual", False):
            raise NotImplementedError("virtual (painted) points: loading.py:16-32 reads side files this package does not")
        nsweeps = res["lidar"]["nsweeps"]
        assert (nsweeps - 1) == len(info["sweeps"]), \
            "nsweeps {} should equal to list length {}.".format(nsweeps, len(info["sweeps"]))
        raws, mats, lags = [read_file(str(info["lidar_path"]))], [None], [0.0]
        for i in np.random.choice(len(info["sweeps"]), nsweeps - 1, replace=False):   # the reference's order
            sweep = info["sweeps"][i]
            raws.append(read_file(str(sweep["lidar_path"])))
            mats.append(sweep["transform_matrix"])
            lags.append(sweep["time_lag"])
        combined = merge_sweeps(raws, mats, lags, self.device)
        res["lidar"]["combined"] = combined
        res["lidar"]["points"] = combined[:, :4]
        res["lidar"]["times"] = combined[:, 4:5]
        return res, info

    def _waymo(self, res, info):
        """loading.py:142-175: the key frame and `nsweeps - 1` earlier frames IN THE LISTED ORDER (no random choice, no
        close-point filter); intensity -> tanh(intensity) on every frame; one sweep: `points` [n, 5] only."""
        nsweeps = res["lidar"]["nsweeps"]
        raws, mats, lags = [_waymo_raw(get_obj(info["path"]))], [None], [0.0]
        if nsweeps > 1:
            assert (nsweeps - 1) == len(info["sweeps"]), \
                "nsweeps {} should be equal to the list length {}.".format(nsweeps, len(info["sweeps"]))
            for i in range(nsweeps - 1):
                sweep = info["sweeps"][i]
                raws.append(_waymo_raw(get_obj(sweep["path"])))
                mats.append(sweep["transform_matrix"])
                lags.append(sweep["time_lag"])
        combined = merge_sweeps(raws, mats, lags, self.device, num_point_feature=5, filter_close=[0] * len(raws),
                                tanh_column=3)
        if nsweeps > 1:
            res["lidar"]["points"] = combined[:, :5]
            res["lidar"]["times"] = combined[:, 5:6]
            res["lidar"]["combined"] = combined
        else:
            res["lidar"]["points"] = combined[:, :5].contiguous()
        return res, info
