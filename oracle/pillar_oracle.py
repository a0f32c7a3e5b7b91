"""CPU restatement (numpy) of the reference's point -> pillar path.  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.

Parity status: **unpinned by the reference** for the index part -- the reference ships no test,
fixture or golden vector for it, and its kernels are CUDA-only.  The
arithmetic below follows the reference source line by line; the float->cell step additionally
restates what `tensor / python_float` does inside torch on a GPU (multiply by the fp32 reciprocal),
because the reference only ever runs there.  `tests/golden/` pins the pieces that *can* be run from
the reference's own Python (DynamicPFE's coordinate/mask step on torch-CPU = `true_divide=True`,
and PillarMaxPooling.shared_mlps).

Follows:
  det3d/models/readers/dynamic_pillar_encoder.py:29-50   (DynamicPFE.forward)
  det3d/ops/pillar_ops/pillar_utils.py:7-58              (bev_spatial_shape, PillarQueryAndGroup)
  det3d/ops/pillar_ops/src/pillar_ops_gpu.cu:13-39,60-78 (index kernels)
  det3d/ops/pillar_ops/src/group_ops_gpu.cu:8-33         (gathers)
  det3d/ops/pillar_ops/pillar_modules.py:26-33,56-74     (Linear+BN1d+ReLU, scatter_max)
  det3d/ops/pillar_ops/src/scatter_ops_gpu.cu:13-22      (scatter_max_kernel, zero-initialised out)
"""
import numpy as np

F32 = np.float32


def bev_spatial_shape(pillar_size, point_cloud_range):
    # pillar_utils.py:7-10
    W = round((point_cloud_range[3] - point_cloud_range[0]) / pillar_size)
    H = round((point_cloud_range[4] - point_cloud_range[1]) / pillar_size)
    return int(H), int(W)


def _float_to_int32_device(v):
    """float32 -> int32 the way the GPU's convert instruction does: saturate, NaN -> 0."""
    v = np.asarray(v, dtype=np.float32)
    out = np.zeros(v.shape, dtype=np.int64)
    finite = ~np.isnan(v)
    clipped = np.clip(v[finite].astype(np.float64), -2147483648.0, 2147483647.0)
    out[finite] = clipped.astype(np.int64)
    return out.astype(np.int32)


def cell_coords(p, pmin, pillar_size, true_divide=False):
    """dynamic_pillar_encoder.py:34-35: ((p - min) / ps).floor().int(), fp32 throughout.

    true_divide=False: torch on a GPU turns `/ python_scalar` into `* (fp32(1)/fp32(ps))`
    (ATen BinaryDivTrueKernel, cpu-scalar fast path) -- the semantics the reference runs with.
    true_divide=True: torch on the CPU divides.
    """
    p = np.asarray(p, dtype=F32)
    v = p - F32(pmin)
    if true_divide:
        v = v / F32(pillar_size)
    else:
        v = v * (F32(1.0) / F32(pillar_size))
    return _float_to_int32_device(np.floor(v))


def dynamic_pfe_select(points_list, pc_range, pillar_size, true_divide=False):
    """DynamicPFE.forward up to the pfn_layers call (dynamic_pillar_encoder.py:30-47).

    Returns pts_xy [L,2] int32 (x, y), pts_batch_cnt [B] int32, pts_features [L,F] fp32."""
    H, W = bev_spatial_shape(pillar_size, pc_range)
    xy, cnt, feats = [], [], []
    for pts in points_list:
        pts = np.asarray(pts, dtype=F32)
        cx = cell_coords(pts[:, 0], pc_range[0], pillar_size, true_divide)
        cy = cell_coords(pts[:, 1], pc_range[1], pillar_size, true_divide)
        m = (cx >= 0) & (cx < W) & (cy >= 0) & (cy < H)
        xy.append(np.stack((cx[m], cy[m]), axis=1))
        feats.append(pts[m])
        cnt.append(int(m.sum()))
    return (np.concatenate(xy).astype(np.int32), np.asarray(cnt, dtype=np.int32),
            np.concatenate(feats).astype(F32))


def pillar_query_and_group(pts_xy, pts_batch_cnt, pts_features, pillar_size, pc_range):
    """PillarQueryAndGroup.forward (pillar_utils.py:22-58).

    flat = b*H*W + y*W + x (pillar_ops_gpu.cu:36); the bool mask + cumsum compaction
    (pillar_utils.py:43-48, pillar_ops_gpu.cu:70-77) lists the occupied cells in ascending flat
    order; gather_indice (group_ops_gpu.cu:16) maps each point to its cell's row."""
    H, W = bev_spatial_shape(pillar_size, pc_range)
    L = pts_xy.shape[0]
    bid = np.repeat(np.arange(len(pts_batch_cnt), dtype=np.int64), pts_batch_cnt)
    assert bid.shape[0] == L
    flat = bid * (H * W) + pts_xy[:, 1].astype(np.int64) * W + pts_xy[:, 0].astype(np.int64)
    uniq, inv = np.unique(flat, return_inverse=True)
    pillar_indices = np.stack((uniq // (H * W), (uniq // W) % H, uniq % W), axis=1).astype(np.int32)
    point_pillar_indices = inv.astype(np.int32)
    # pillar_utils.py:19-20,51-56 -- int32 * python float -> fp32 mul, then fp32 add of the offset
    x_off = F32(pillar_size / 2.0 + pc_range[0])
    y_off = F32(pillar_size / 2.0 + pc_range[1])
    ctr_x = pillar_indices[:, 2].astype(F32) * F32(pillar_size) + x_off
    ctr_y = pillar_indices[:, 1].astype(F32) * F32(pillar_size) + y_off
    centers = np.stack((ctr_x, ctr_y), axis=1)[point_pillar_indices]
    rel = pts_features[:, :2].astype(F32) - centers
    feats = np.concatenate((rel, pts_features.astype(F32)), axis=1)
    return pillar_indices, point_pillar_indices, feats


def fold_bn(gamma, beta, mean, var, eps):
    """eval-mode BatchNorm as y = x*scale + shift (fp32)."""
    scale = (np.asarray(gamma, np.float64) / np.sqrt(np.asarray(var, np.float64) + eps))
    shift = np.asarray(beta, np.float64) - np.asarray(mean, np.float64) * scale
    return scale.astype(F32), shift.astype(F32)


def pfn_mlp(feats, weight, gamma, beta, mean, var, eps=1e-3):
    """pillar_modules.py:26-33 in eval mode: relu(BN1d(x @ W^T)).  fp64 accumulate -> fp32."""
    x = feats.astype(np.float64) @ np.asarray(weight, np.float64).T
    x = (x - np.asarray(mean, np.float64)) / np.sqrt(np.asarray(var, np.float64) + eps)
    x = x * np.asarray(gamma, np.float64) + np.asarray(beta, np.float64)
    return np.maximum(x, 0.0).astype(F32)


def scatter_max(src, index, M):
    """scatter_utils.py:7-27 + scatter_ops_gpu.cu:13-22: out starts at 0, out[index[p]] = max(...)."""
    out = np.zeros((M, src.shape[1]), dtype=F32)
    np.maximum.at(out, index, src)
    return out


def pillarize(points_list, pc_range, pillar_size, weight, gamma, beta, mean, var, eps=1e-3,
              true_divide=False):
    """The whole reader: returns dict(indices [M,3], features [M,C], point_pillar_index per
    *input* point (-1 when out of range), feats7 [L,F+2])."""
    pts_xy, cnt, pf = dynamic_pfe_select(points_list, pc_range, pillar_size, true_divide)
    idx, ppi, feats = pillar_query_and_group(pts_xy, cnt, pf, pillar_size, pc_range)
    y = pfn_mlp(feats, weight, gamma, beta, mean, var, eps)
    out = scatter_max(y, ppi, idx.shape[0])
    # per-input-point map (the product returns it for every input row)
    H, W = bev_spatial_shape(pillar_size, pc_range)
    full = []
    k = 0
    for pts in points_list:
        pts = np.asarray(pts, dtype=F32)
        cx = cell_coords(pts[:, 0], pc_range[0], pillar_size, true_divide)
        cy = cell_coords(pts[:, 1], pc_range[1], pillar_size, true_divide)
        m = (cx >= 0) & (cx < W) & (cy >= 0) & (cy < H)
        f = np.full(pts.shape[0], -1, dtype=np.int32)
        n = int(m.sum())
        f[m] = ppi[k:k + n]
        k += n
        full.append(f)
    return dict(indices=idx, features=out, point_pillar_index=np.concatenate(full), feats7=feats,
                point_feats_out=y)
