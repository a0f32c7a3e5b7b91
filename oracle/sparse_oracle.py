"""CPU restatement of the sparse-conv semantics the backbone relies on.  TEST INFRASTRUCTURE ONLY.

**Parity unpinned**: the arithmetic lives in `spconv-cu114` (PyPI, version not pinned by the
reference: docs/requirements.txt:29), which is neither vendored under /root/reference nor
installed here, and the reference holds no test or golden vector at this boundary.  What is
restated is spconv 2.x's published definition, anchored on the reference's call sites
(det3d/models/backbones/base.py:38-63, PillarResNet.py:81-108):

  SubMConv2d(3x3, padding=1):  out[site] = bias + sum_{ky,kx} W[:,ky,kx,:] . x[site + (ky-1, kx-1)]
      over ACTIVE neighbours only; the output's active set is the input's.
  SparseConv2d(3x3, stride 2, padding 1): output grid floor((H+2-3)/2)+1; an output cell is active
      iff its 3x3 window holds an active input; value = sum over the active inputs (+ bias).
  Both equal F.conv2d(zero-filled dense map) masked to the active outputs -- `dense_definition`
  below states that directly and tests/test_oracle_cpu.py checks the gather form against it.
  Output rows are listed in ascending flat index (spconv's own order is implementation-defined;
  dense maps are what is compared).
"""
import numpy as np

F32 = np.float32


def _flat(idx, H, W):
    idx = idx.astype(np.int64)
    return idx[:, 0] * (H * W) + idx[:, 1] * W + idx[:, 2]


def subm_conv2d(feats, indices, spatial_shape, weight, bias=None):
    """feats [M,Cin], indices [M,3] (b,y,x) sorted by flat index, weight [Cout,3,3,Cin]."""
    H, W = spatial_shape
    flat = _flat(indices, H, W)
    assert np.all(np.diff(flat) > 0)
    M = feats.shape[0]
    out = np.zeros((M, weight.shape[0]), dtype=np.float64)
    x = feats.astype(np.float64)
    for ky in range(3):
        for kx in range(3):
            ny = indices[:, 1].astype(np.int64) + ky - 1
            nx = indices[:, 2].astype(np.int64) + kx - 1
            ok = (ny >= 0) & (ny < H) & (nx >= 0) & (nx < W)
            nflat = indices[:, 0].astype(np.int64) * (H * W) + ny * W + nx
            pos = np.searchsorted(flat, nflat)
            pos[pos >= M] = M - 1 if M else 0
            hit = ok & (flat[pos] == nflat) if M else ok
            rows = np.nonzero(hit)[0]
            out[rows] += x[pos[rows]] @ weight[:, ky, kx, :].astype(np.float64).T
    if bias is not None:
        out += np.asarray(bias, np.float64)
    return out.astype(F32)


def sparse_conv2d(feats, indices, spatial_shape, batch, weight, bias=None, stride=2, pad=1):
    """Returns (out_feats [Mo,Cout], out_indices [Mo,3], (Ho,Wo))."""
    H, W = spatial_shape
    Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    flat = _flat(indices, H, W)
    b = indices[:, 0].astype(np.int64)
    y = indices[:, 1].astype(np.int64)
    x = indices[:, 2].astype(np.int64)
    cand = []
    for ky in range(3):
        for kx in range(3):
            ty, tx = y + pad - ky, x + pad - kx
            ok = (ty >= 0) & (tx >= 0) & (ty % stride == 0) & (tx % stride == 0)
            oy, ox = ty // stride, tx // stride
            ok &= (oy < Ho) & (ox < Wo)
            cand.append((b * Ho * Wo + oy * Wo + ox)[ok])
    oflat = np.unique(np.concatenate(cand)) if len(cand) else np.zeros(0, np.int64)
    out_idx = np.stack((oflat // (Ho * Wo), (oflat // Wo) % Ho, oflat % Wo), axis=1).astype(np.int32)
    Mo, M = oflat.shape[0], feats.shape[0]
    out = np.zeros((Mo, weight.shape[0]), dtype=np.float64)
    xin = feats.astype(np.float64)
    ob, oy, ox = out_idx[:, 0].astype(np.int64), out_idx[:, 1].astype(np.int64), out_idx[:, 2].astype(np.int64)
    for ky in range(3):
        for kx in range(3):
            iy, ix = oy * stride - pad + ky, ox * stride - pad + kx
            ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
            nflat = ob * (H * W) + iy * W + ix
            pos = np.searchsorted(flat, nflat)
            pos[pos >= M] = M - 1 if M else 0
            hit = ok & (flat[pos] == nflat) if M else ok & False
            rows = np.nonzero(hit)[0]
            out[rows] += xin[pos[rows]] @ weight[:, ky, kx, :].astype(np.float64).T
    if bias is not None:
        out += np.asarray(bias, np.float64)
    return out.astype(F32), out_idx, (Ho, Wo)


def batchnorm_eval(x, gamma, beta, mean, var, eps=1e-3):
    x = (x.astype(np.float64) - np.asarray(mean, np.float64)) / np.sqrt(np.asarray(var, np.float64) + eps)
    return (x * np.asarray(gamma, np.float64) + np.asarray(beta, np.float64)).astype(F32)


def to_dense(feats, indices, spatial_shape, batch):
    H, W = spatial_shape
    out = np.zeros((batch, feats.shape[1], H, W), dtype=feats.dtype)
    out[indices[:, 0], :, indices[:, 1], indices[:, 2]] = feats
    return out


def dense_definition(feats, indices, spatial_shape, batch, weight, bias, subm, stride=2, pad=1):
    """The same two ops written as a masked dense torch conv (definition used for cross-checks).
    Returns the dense [B,Cout,Ho,Wo] map (zero at inactive outputs)."""
    import torch
    import torch.nn.functional as Fn
    H, W = spatial_shape
    dense = torch.from_numpy(to_dense(feats.astype(F32), indices, spatial_shape, batch)).double()
    mask = torch.zeros((batch, 1, H, W), dtype=torch.float64)
    mask[indices[:, 0], 0, indices[:, 1], indices[:, 2]] = 1.0
    w = torch.from_numpy(np.ascontiguousarray(weight)).double().permute(0, 3, 1, 2)
    if subm:
        y = Fn.conv2d(dense, w, None, 1, 1)
        mo = mask
    else:
        y = Fn.conv2d(dense, w, None, stride, pad)
        mo = (Fn.max_pool2d(mask, 3, stride, pad) > 0).double()
    if bias is not None:
        y = y + torch.from_numpy(np.asarray(bias, np.float64)).view(1, -1, 1, 1)
    return (y * mo).float().numpy(), mo.numpy()
