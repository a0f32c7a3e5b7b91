// Operators of the training / unfused path, with the reference's pybind signatures
// (det3d/ops/pillar_ops/src/pillar_api.cpp:10-21): point->pillar indexing, gathers, scatter-max
// with argmax and its gradient, plus the inverse rulebook the sparse-conv data gradient needs.
//
//   pn_create_point_pillar_index_stack  pillar_ops_gpu.cu:13-58   (bool mask + per-point flat cell)
//   pn_create_pillar_indices            pillar_ops_gpu.cu:60-94   (position map -> (b,y,x) rows)
//   pn_gather_indice / pn_gather_feature  group_ops_gpu.cu:8-78
//   pn_scatter_max / pn_scatter_max_grad  scatter_ops_gpu.cu:13-75
//
// scatter-max here is ONE pass with a native 64-bit atomic max on (value bits, ~point index): the
// value part reproduces `out = max(0, max_p src[p])` (out starts at 0, scatter_utils.py:20) and the
// index part makes `arg` deterministic -- the lowest point index among the maxima -- where the
// reference's second pass (`abs(src - out) < 1e-5`, last writer wins) is order-dependent.
#include "common.h"

static constexpr int T = 256;
static int blocks_for(long long n) {
  long long b = (n + T - 1) / T;
  return (int)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

__global__ __launch_bounds__(T) void point_pillar_index_kernel(int N, int B, int H, int W,
                                                               const int32_t* __restrict__ pts_xy,
                                                               const int32_t* __restrict__ cnt,
                                                               unsigned char* __restrict__ mask,
                                                               int32_t* __restrict__ ppi) {
  for (int i = blockIdx.x * T + threadIdx.x; i < N; i += gridDim.x * T) {
    int b = 0, acc = cnt[0];
    for (int k = 1; k < B; ++k) {
      if (i < acc) break;
      acc += cnt[k];
      b = k;
    }
    const int x = pts_xy[2 * i], y = pts_xy[2 * i + 1];
    if (x < 0 || x >= W || y < 0 || y >= H) continue;
    const int idx = (b * H + y) * W + x;
    mask[idx] = 1;
    ppi[i] = idx;
  }
}

__global__ __launch_bounds__(T) void pillar_indices_kernel(long long cells, int H, int W,
                                                           const int32_t* __restrict__ pos,
                                                           int32_t* __restrict__ out) {
  for (long long c = (long long)blockIdx.x * T + threadIdx.x; c < cells; c += (long long)gridDim.x * T) {
    const int p = pos[c];
    if (p < 0) continue;
    const int hw = H * W;
    const int b = (int)(c / hw), rem = (int)(c - (long long)b * hw), y = rem / W;
    out[p * 3 + 0] = b;
    out[p * 3 + 1] = y;
    out[p * 3 + 2] = rem - y * W;
  }
}

__global__ __launch_bounds__(T) void gather_indice_kernel(int L, const int32_t* __restrict__ index,
                                                          const int32_t* __restrict__ table,
                                                          int32_t* __restrict__ out) {
  for (int i = blockIdx.x * T + threadIdx.x; i < L; i += gridDim.x * T) out[i] = table[index[i]];
}

__global__ __launch_bounds__(T) void gather_feature_kernel(long long total, int C,
                                                           const int32_t* __restrict__ index,
                                                           const float* __restrict__ feats,
                                                           float* __restrict__ out) {
  for (long long t = (long long)blockIdx.x * T + threadIdx.x; t < total; t += (long long)gridDim.x * T) {
    const long long i = t / C;
    const int c = (int)(t - i * C);
    out[t] = feats[(size_t)index[i] * C + c];
  }
}

// gather_feature backward (group_ops_gpu.cu:35-48): grad_features[index[l]] += grad_outs[l]; one lane per
// (l, c) so the adds of a row are coalesced (the reference walks a row's channels in one thread)
__global__ __launch_bounds__(T) void gather_feature_grad_kernel(long long total, int C,
                                                                const int32_t* __restrict__ index,
                                                                const float* __restrict__ grad_outs,
                                                                float* __restrict__ grad_feats) {
  for (long long t = (long long)blockIdx.x * T + threadIdx.x; t < total; t += (long long)gridDim.x * T) {
    const long long i = t / C;
    const int c = (int)(t - i * C);
    atomicAdd(grad_feats + (size_t)index[i] * C + c, grad_outs[t]);
  }
}

// ---- scatter max -------------------------------------------------------------------------------
__global__ __launch_bounds__(T) void scatter_max_pack_kernel(long long total, int C,
                                                             const int32_t* __restrict__ index,
                                                             const float* __restrict__ src,
                                                             unsigned long long* __restrict__ packed) {
  for (long long t = (long long)blockIdx.x * T + threadIdx.x; t < total; t += (long long)gridDim.x * T) {
    const long long p = t / C;
    const int c = (int)(t - p * C);
    const float v = src[t];
    if (!(v > 0.f)) continue;   // cannot beat the zero the output starts from
    const unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) |
                                   (unsigned long long)(0xFFFFFFFFu - (unsigned)p);
    unsigned long long* slot = packed + (size_t)index[p] * C + c;
    if (key > *slot) atomicMax(slot, key);
  }
}

__global__ __launch_bounds__(T) void scatter_max_unpack_kernel(long long total,
                                                               const unsigned long long* __restrict__ packed,
                                                               float* __restrict__ out, int32_t* __restrict__ arg) {
  for (long long t = (long long)blockIdx.x * T + threadIdx.x; t < total; t += (long long)gridDim.x * T) {
    const unsigned long long k = packed[t];
    out[t] = k ? __uint_as_float((unsigned)(k >> 32)) : 0.f;
    if (arg) arg[t] = k ? (int32_t)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull)) : -1;
  }
}

__global__ __launch_bounds__(T) void scatter_max_grad_kernel(long long total, int C,
                                                             const float* __restrict__ grad_out,
                                                             const int32_t* __restrict__ arg,
                                                             float* __restrict__ grad_src) {
  for (long long t = (long long)blockIdx.x * T + threadIdx.x; t < total; t += (long long)gridDim.x * T) {
    const int p = arg[t];
    if (p < 0) continue;
    const int c = (int)(t % C);
    grad_src[(size_t)p * C + c] = grad_out[t];   // one writer per (point, channel): arg is unique per (m, c)
  }
}

// inverse of a rulebook: inv[k][j] = output row m with nbr[k][m] == j (at most one), else -1
__global__ __launch_bounds__(T) void invert_rules_kernel(const int32_t* __restrict__ nbr, int nbr_stride,
                                                         const int32_t* __restrict__ out_count, int out_cap,
                                                         int32_t* __restrict__ inv, int inv_stride) {
  int M = *out_count;
  if (M > out_cap) M = out_cap;
  const long long total = 9ll * M;
  for (long long t = (long long)blockIdx.x * T + threadIdx.x; t < total; t += (long long)gridDim.x * T) {
    const int k = (int)(t / M), m = (int)(t - (long long)k * M);
    const int j = nbr[(size_t)k * nbr_stride + m];
    if (j >= 0) inv[(size_t)k * inv_stride + j] = m;
  }
}

#define LAUNCH(kern, n, ...)                                                                   \
  hipLaunchKernelGGL(kern, dim3(blocks_for(n)), dim3(T), 0, stream, __VA_ARGS__);              \
  PN_CHECK_LAUNCH(#kern)

extern "C" int pn_create_point_pillar_index_stack(const int32_t* pts_xy, const int32_t* pts_batch_cnt, int N,
                                                  int B, int H, int W, uint8_t* pillars_mask,
                                                  int32_t* point_pillar_index, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(pts_batch_cnt && pillars_mask && (N == 0 || (pts_xy && point_pillar_index)) && B >= 1,
               "pn_create_point_pillar_index_stack: null argument");
  if (N == 0) return PN_OK;
  LAUNCH(point_pillar_index_kernel, N, N, B, H, W, pts_xy, pts_batch_cnt, pillars_mask, point_pillar_index);
  return PN_OK;
}

extern "C" int pn_create_pillar_indices(const int32_t* pillars_position, int B, int H, int W,
                                        int32_t* pillar_indices, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(pillars_position && pillar_indices, "pn_create_pillar_indices: null argument");
  const long long cells = (long long)B * H * W;
  LAUNCH(pillar_indices_kernel, cells, cells, H, W, pillars_position, pillar_indices);
  return PN_OK;
}

extern "C" int pn_gather_indice(const int32_t* index, int L, const int32_t* indices, int32_t* outs,
                                pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (L == 0) return PN_OK;
  PN_CHECK_ARG(index && indices && outs, "pn_gather_indice: null argument");
  LAUNCH(gather_indice_kernel, L, L, index, indices, outs);
  return PN_OK;
}

extern "C" int pn_gather_feature(const int32_t* index, int L, const float* features, int C, float* outs,
                                 pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (L == 0) return PN_OK;
  PN_CHECK_ARG(index && features && outs && C >= 1, "pn_gather_feature: null argument");
  LAUNCH(gather_feature_kernel, (long long)L * C, (long long)L * C, C, index, features, outs);
  return PN_OK;
}

extern "C" int pn_gather_feature_grad(const float* grad_outs, const int32_t* index, int L, int C,
                                      float* grad_features, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (L == 0) return PN_OK;
  PN_CHECK_ARG(index && grad_outs && grad_features && C >= 1, "pn_gather_feature_grad: null argument");
  LAUNCH(gather_feature_grad_kernel, (long long)L * C, (long long)L * C, C, index, grad_outs, grad_features);
  return PN_OK;
}

extern "C" size_t pn_scatter_max_workspace_bytes(int M, int C) { return pn_align_up((size_t)M * C * 8, 256); }

extern "C" int pn_scatter_max(const int32_t* index, const float* src, int L, int C, int M, int32_t* arg,
                              float* out, void* workspace, size_t workspace_bytes, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(out && workspace && C >= 1 && M >= 0 && L >= 0, "pn_scatter_max: null argument");
  if (M == 0) return PN_OK;
  if (workspace_bytes < pn_scatter_max_workspace_bytes(M, C)) {
    pn_set_error("pn_scatter_max: workspace too small");
    return PN_ENOSPC;
  }
  unsigned long long* packed = (unsigned long long*)workspace;
  { const int rc_ = pn_fill_async(packed, 0, (size_t)M * C * 8, stream); if (rc_ != PN_OK) return rc_; }
  if (L > 0) {
    PN_CHECK_ARG(index && src, "pn_scatter_max: null input");
    LAUNCH(scatter_max_pack_kernel, (long long)L * C, (long long)L * C, C, index, src, packed);
  }
  LAUNCH(scatter_max_unpack_kernel, (long long)M * C, (long long)M * C, packed, out, arg);
  return PN_OK;
}

extern "C" int pn_scatter_max_grad(const float* grad_out, const int32_t* arg, int M, int C, float* grad_src,
                                   pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M == 0) return PN_OK;
  PN_CHECK_ARG(grad_out && arg && grad_src, "pn_scatter_max_grad: null argument");
  LAUNCH(scatter_max_grad_kernel, (long long)M * C, (long long)M * C, C, grad_out, arg, grad_src);
  return PN_OK;
}

extern "C" int pn_spconv_invert_rules(const int32_t* nbr, int nbr_stride, const int32_t* out_count_dev,
                                      int out_cap, int32_t* inv, int inv_stride, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(nbr && out_count_dev && inv && out_cap >= 1 && inv_stride >= 1, "pn_spconv_invert_rules: bad argument");
  { const int rc_ = pn_fill_async(inv, 0xFF, (size_t)9 * inv_stride * 4, stream); if (rc_ != PN_OK) return rc_; }
  LAUNCH(invert_rules_kernel, 9ll * out_cap, nbr, nbr_stride, out_count_dev, out_cap, inv, inv_stride);
  return PN_OK;
}
