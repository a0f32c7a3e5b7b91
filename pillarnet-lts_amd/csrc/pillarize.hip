// (a1-a4) raw points -> sorted pillars + max-pooled pillar features, gfx950.
//
// Reference path being replaced (one call instead of ~25 launches, a dense
// bool mask, two 8.3 MB cumsum passes, a `.item()` sync and a CAS-loop float
// atomMax):
//   det3d/models/readers/dynamic_pillar_encoder.py:29-50   cell coords + range mask
//   det3d/ops/pillar_ops/pillar_utils.py:22-58             index / group
//   det3d/ops/pillar_ops/pillar_modules.py:56-74           Linear+BN+ReLU, scatter_max
//   det3d/ops/pillar_ops/src/{pillar_ops,group_ops,scatter_ops}_gpu.cu
//
// Launches: memset(byte map) -> mark -> 2 x occupancy build -> pfn_scatter.
//   mark:        one pass over the point rows; cell id per point and a plain byte
//                store into the cell's occupancy byte (same value from every writer:
//                no atomics; the 64-bit atomic OR version measured 30 us for 300k
//                points, 10x this).
//   scan:        occupancy.hip (rank per word, (b,y,x) rows, count, zero fill).
//   pfn_scatter: half a wavefront per point (lane = output channel): 7 FMAs,
//                folded BN, ReLU, then a *native* integer atomic max on the
//                fp32 bit pattern (values are >= 0, so int order == float
//                order) guarded by a plain read -- a wave-instruction touches
//                two full 128-B rows, the shape global atomics run fastest at.
//                The result is order-independent, hence deterministic.
#include "common.h"

struct BatchOffsets {
  int32_t off[PN_MAX_BATCH + 1];
};

struct GridDev {
  int B, H, W;
  float x_min, y_min, inv_pillar, pillar_size, x_offset, y_offset;
  int true_divide;
};

static constexpr int PZ_THREADS = 256;

// torch semantics of `((p - min) / ps).floor().int()` on the device:
// sub, then mul by the fp32 reciprocal (BinaryDivTrueKernel's cpu-scalar fast
// path), floor, saturating cvt (NaN -> 0).  -ffp-contract=off keeps the ops
// separately rounded as the separate torch kernels are.
__device__ __forceinline__ int cell_coord(float p, float pmin, float inv, float ps, int true_div) {
  float v = p - pmin;
  v = true_div ? v / ps : v * inv;
  return (int)floorf(v);
}

template <int F>
__global__ __launch_bounds__(PZ_THREADS) void mark_kernel(const float* __restrict__ pts, int n,
                                                          GridDev g, BatchOffsets bo,
                                                          int32_t* __restrict__ flat,
                                                          unsigned char* __restrict__ occ_bytes) {
  const int stride = gridDim.x * PZ_THREADS;
  for (int base = blockIdx.x * PZ_THREADS; base < n; base += stride) {
    const int i = base + threadIdx.x;
    int idx = -1;
    if (i < n) {
      const float x = pts[(size_t)i * F + 0];
      const float y = pts[(size_t)i * F + 1];
      const int cx = cell_coord(x, g.x_min, g.inv_pillar, g.pillar_size, g.true_divide);
      const int cy = cell_coord(y, g.y_min, g.inv_pillar, g.pillar_size, g.true_divide);
      if (cx >= 0 && cx < g.W && cy >= 0 && cy < g.H) {
        int b = 0;  // pillar_ops_gpu.cu:24-29 scans pts_batch_cnt the same way
        while (b + 1 < g.B && i >= bo.off[b + 1]) ++b;
        idx = (b * g.H + cy) * g.W + cx;
      }
      flat[i] = idx;
      if (idx >= 0) occ_bytes[idx] = 1;
    }
  }
}

template <int F, int C>
__global__ __launch_bounds__(PZ_THREADS) void pfn_scatter_kernel(
    const float* __restrict__ pts, int n, GridDev g, const int32_t* __restrict__ flat,
    const unsigned long long* __restrict__ bits, const int32_t* __restrict__ word_rank,
    const float* __restrict__ weight /*[C][F+2]*/, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out /*[M][C]*/,
    int32_t* __restrict__ point_pillar_index) {
  static_assert(C == 32, "lane = channel inside a half wavefront");
  const int lane = threadIdx.x & 63;
  const int ch = lane & 31, half = lane >> 5;
  float w[F + 2];
#pragma unroll
  for (int k = 0; k < F + 2; ++k) w[k] = weight[ch * (F + 2) + k];
  const float sc = scale[ch], sh = shift[ch];

  const int stride = gridDim.x * PZ_THREADS;
  for (int base = blockIdx.x * PZ_THREADS; base < n; base += stride) {
    const int i = base + threadIdx.x;
    float f[F + 2];
#pragma unroll
    for (int k = 0; k < F + 2; ++k) f[k] = 0.f;
    int pid = -1;
    if (i < n) {
      const int idx = flat[i];
      if (idx >= 0) {
        pid = pn_occ_rank(bits, word_rank, idx);
#pragma unroll
        for (int k = 0; k < F; ++k) f[2 + k] = pts[(size_t)i * F + k];
        const int cx = idx % g.W, cy = (idx / g.W) % g.H;
        // pillar_utils.py:51-54: centre = idx * ps + (ps/2 + min), offset = xy - centre
        const float ctr_x = (float)cx * g.pillar_size + g.x_offset;
        const float ctr_y = (float)cy * g.pillar_size + g.y_offset;
        f[0] = f[2] - ctr_x;
        f[1] = f[3] - ctr_y;
      }
      if (point_pillar_index) point_pillar_index[i] = pid;
    }
    // 32 steps, two points per step: half h of the wave takes point 2*s + h
#pragma unroll 4
    for (int s = 0; s < 32; ++s) {
      const int src = 2 * s + half;
      const int p = __shfl(pid, src);
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < F + 2; ++k) acc = fmaf(__shfl(f[k], src), w[k], acc);
      float v = acc * sc + sh;
      v = fmaxf(v, 0.f);
      if (p >= 0 && v > 0.f) {
        float* o = out + (size_t)p * C + ch;
        if (v > *o) atomicMax(reinterpret_cast<int*>(o), __float_as_int(v));
      }
    }
  }
}

extern "C" size_t pn_pillarize_workspace_bytes(const pnPillarGrid* g, int64_t n_points) {
  if (!g || n_points < 0) return 0;
  const int64_t cells = (int64_t)g->batch * g->height * g->width;
  const pnScanGeom sg = pn_scan_geom(cells);
  size_t b = 0;
  b += pn_align_up((size_t)sg.words * 8, 256);        // bits
  b += pn_align_up((size_t)sg.words * 4, 256);        // word_rank
  b += pn_align_up((size_t)sg.blocks * 4, 256);       // block sums
  b += pn_align_up((size_t)(n_points > 0 ? n_points : 1) * 4, 256);  // flat cell id per point
  b += pn_align_up((size_t)cells, 256);                               // occupancy byte map
  return b;
}

extern "C" int pn_pillarize(const pnPillarGrid* g, const float* points, int64_t n_points,
                            const int32_t* batch_offsets_host, const float* pfn_weight,
                            const float* pfn_scale, const float* pfn_shift,
                            int32_t* pillar_indices, float* pillar_features,
                            int32_t* point_pillar_index, int32_t* num_pillars, void* workspace,
                            size_t workspace_bytes, uint64_t** occ_bits, int32_t** occ_rank,
                            pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(g && batch_offsets_host && pfn_weight && pfn_scale && pfn_shift && pillar_indices &&
                   pillar_features && num_pillars && workspace,
               "pn_pillarize: null argument");
  PN_CHECK_ARG(g->batch >= 1 && g->batch <= PN_MAX_BATCH, "pn_pillarize: batch %d not in 1..%d",
               g->batch, PN_MAX_BATCH);
  PN_CHECK_ARG(g->height > 0 && g->width > 0, "pn_pillarize: bad grid %dx%d", g->height, g->width);
  PN_CHECK_ARG(g->out_channels == 32, "pn_pillarize: out_channels %d (only 32 is built)",
               g->out_channels);
  PN_CHECK_ARG(g->point_feats >= 4 && g->point_feats <= 6,
               "pn_pillarize: point_feats %d (4..6 are built)", g->point_feats);
  PN_CHECK_ARG(n_points >= 0 && n_points < (1ll << 31), "pn_pillarize: n_points %lld",
               (long long)n_points);
  const int64_t cells = (int64_t)g->batch * g->height * g->width;
  PN_CHECK_ARG(cells < (1ll << 31), "pn_pillarize: %lld cells overflow int32", (long long)cells);
  PN_CHECK_ARG(n_points == 0 || points, "pn_pillarize: null points");
  PN_CHECK_ARG(batch_offsets_host[0] == 0 && batch_offsets_host[g->batch] == n_points,
               "pn_pillarize: batch offsets must run 0..n_points");
  for (int b = 0; b < g->batch; ++b)
    PN_CHECK_ARG(batch_offsets_host[b] <= batch_offsets_host[b + 1],
                 "pn_pillarize: batch offsets not ascending");
  if (workspace_bytes < pn_pillarize_workspace_bytes(g, n_points)) {
    pn_set_error("pn_pillarize: workspace %zu < %zu bytes", workspace_bytes,
                 pn_pillarize_workspace_bytes(g, n_points));
    return PN_ENOSPC;
  }
  PN_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)pillar_features & 15) == 0,
               "pn_pillarize: workspace must be 256-B and features 16-B aligned");

  const pnScanGeom sg = pn_scan_geom(cells);
  char* ws = (char*)workspace;
  unsigned long long* bits = (unsigned long long*)ws;
  ws += pn_align_up((size_t)sg.words * 8, 256);
  int32_t* word_rank = (int32_t*)ws;
  ws += pn_align_up((size_t)sg.words * 4, 256);
  int32_t* block_sums = (int32_t*)ws;
  ws += pn_align_up((size_t)sg.blocks * 4, 256);
  int32_t* flat = (int32_t*)ws;
  ws += pn_align_up((size_t)(n_points > 0 ? n_points : 1) * 4, 256);
  unsigned char* occ_bytes = (unsigned char*)ws;
  if (occ_bits) *occ_bits = (uint64_t*)bits;
  if (occ_rank) *occ_rank = word_rank;

  GridDev gd;
  gd.B = g->batch; gd.H = g->height; gd.W = g->width;
  gd.x_min = g->x_min; gd.y_min = g->y_min; gd.inv_pillar = g->inv_pillar;
  gd.pillar_size = g->pillar_size; gd.x_offset = g->x_offset; gd.y_offset = g->y_offset;
  gd.true_divide = g->true_divide;
  BatchOffsets bo;
  for (int b = 0; b <= PN_MAX_BATCH; ++b)
    bo.off[b] = batch_offsets_host[b <= g->batch ? b : g->batch];

  const int64_t row_cap = n_points < cells ? n_points : cells;
  { const int rc_ = pn_fill_async(occ_bytes, 0, (size_t)cells, stream); if (rc_ != PN_OK) return rc_; }
  const int n = (int)n_points;
  const int pt_blocks = n > 0 ? (pn_div_up(n, PZ_THREADS) < 2048 ? pn_div_up(n, PZ_THREADS) : 2048) : 0;

#define PN_DISPATCH_F(F_)                                                                          \
  case F_:                                                                                         \
    if (pt_blocks)                                                                                 \
      hipLaunchKernelGGL((mark_kernel<F_>), dim3(pt_blocks), dim3(PZ_THREADS), 0, stream, points,  \
                         n, gd, bo, flat, occ_bytes);                                                   \
    break;
  switch (g->point_feats) { PN_DISPATCH_F(4) PN_DISPATCH_F(5) PN_DISPATCH_F(6) }
#undef PN_DISPATCH_F
  PN_CHECK_LAUNCH("mark_kernel");

  int rc = pn_occupancy_from_bytes(occ_bytes, bits, cells, g->height, g->width, word_rank, block_sums,
                                   pillar_indices, num_pillars, pillar_features,
                                   g->out_channels * (int)sizeof(float), row_cap, stream);
  if (rc != PN_OK) return rc;

#define PN_DISPATCH_F(F_)                                                                          \
  case F_:                                                                                         \
    if (pt_blocks)                                                                                 \
      hipLaunchKernelGGL((pfn_scatter_kernel<F_, 32>), dim3(pt_blocks), dim3(PZ_THREADS), 0,       \
                         stream, points, n, gd, flat, bits, word_rank, pfn_weight, pfn_scale,      \
                         pfn_shift, pillar_features, point_pillar_index);                          \
    break;
  switch (g->point_feats) { PN_DISPATCH_F(4) PN_DISPATCH_F(5) PN_DISPATCH_F(6) }
#undef PN_DISPATCH_F
  PN_CHECK_LAUNCH("pfn_scatter_kernel");
  return PN_OK;
}
