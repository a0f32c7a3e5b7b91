// Multi-sweep point-cloud merge on the device: the nuScenes branch of LoadPointCloudFromFile
// (det3d/datasets/pipelines/loading.py:103-140) with read_sweep / remove_close (:38-61), and its Waymo branch
// (:142-175 with read_single_waymo / read_single_waymo_sweep, :63-94: no close-point filter, tanh of the intensity
// column of every sweep) --
//   key sweep as it is (dt = 0); every other sweep: drop points with |x| < r AND |y| < r (r = 1 m, tested on
//   the raw sensor-frame coordinates), move xyz by the sweep's 4x4 transform (float64 arithmetic, rounded
//   to fp32 on store, as numpy's float64 dot assigned into a float32 array), append the sweep's time lag.
// Output rows keep the reference's order (sweeps in the given order, points in file order): a stable
// compaction -- per-block counts, one scan block, ballot-ranked writes -- not an atomic append.
#include "common.h"

static constexpr int SW_T = 256;
static constexpr int SW_MAX = 64;   // sweeps per call

struct SweepMeta {
  int n_sweeps;
  int start[SW_MAX + 1];    // first raw row of each sweep (start[n] = total)
};

__device__ __forceinline__ bool sweep_keep(const float* __restrict__ raw, int raw_dim, long long i, int s,
                                           const uint8_t* __restrict__ filter_close, float radius) {
  if (!filter_close[s]) return true;
  const float x = raw[i * raw_dim], y = raw[i * raw_dim + 1];
  return !(fabsf(x) < radius && fabsf(y) < radius);
}

__device__ __forceinline__ int sweep_of(const SweepMeta& M, int i) {
  int s = 0;
  while (s + 1 < M.n_sweeps && i >= M.start[s + 1]) ++s;
  return s;
}

__global__ __launch_bounds__(SW_T) void sweeps_count_kernel(SweepMeta M, const float* __restrict__ raw, int raw_dim,
                                                            const uint8_t* __restrict__ filter_close, float radius,
                                                            int32_t* __restrict__ block_sums) {
  __shared__ int wsum[SW_T / 64];
  const int i = blockIdx.x * SW_T + threadIdx.x, total = M.start[M.n_sweeps];
  const bool keep = i < total && sweep_keep(raw, raw_dim, i, sweep_of(M, i), filter_close, radius);
  const unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    for (int w = 0; w < SW_T / 64; ++w) c += wsum[w];
    block_sums[blockIdx.x] = c;
  }
}

__global__ __launch_bounds__(1024) void sweeps_scan_kernel(int32_t* __restrict__ block_sums, int nblocks,
                                                           int32_t* __restrict__ out_count) {
  __shared__ int part[1024];
  const int per = (nblocks + 1023) / 1024, lo = threadIdx.x * per;
  int s = 0;
  for (int k = 0; k < per; ++k)
    if (lo + k < nblocks) s += block_sums[lo + k];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - s;   // exclusive prefix of this thread's chunk
  for (int k = 0; k < per; ++k)
    if (lo + k < nblocks) {
      const int c = block_sums[lo + k];
      block_sums[lo + k] = run;
      run += c;
    }
  if (threadIdx.x == 1023) *out_count = part[1023];
}

__global__ __launch_bounds__(SW_T) void sweeps_write_kernel(SweepMeta M, const float* __restrict__ raw, int raw_dim,
                                                            int use_dim, const double* __restrict__ transform,
                                                            const uint8_t* __restrict__ has_transform,
                                                            const float* __restrict__ time_lag,
                                                            const uint8_t* __restrict__ filter_close, float radius,
                                                            const int32_t* __restrict__ block_offsets,
                                                            int tanh_column, float* __restrict__ out) {
  __shared__ int wsum[SW_T / 64];
  const int i = blockIdx.x * SW_T + threadIdx.x, total = M.start[M.n_sweeps];
  const int s = i < total ? sweep_of(M, i) : 0;
  const bool keep = i < total && sweep_keep(raw, raw_dim, i, s, filter_close, radius);
  const unsigned long long m = __ballot(keep);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wsum[wv] = __popcll(m);
  __syncthreads();
  if (!keep) return;
  int pos = block_offsets[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
  for (int w = 0; w < wv; ++w) pos += wsum[w];
  const float* p = raw + (long long)i * raw_dim;
  float* o = out + (long long)pos * (use_dim + 1);
  float x = p[0], y = p[1], z = p[2];
  if (has_transform[s]) {
    const double* T = transform + s * 16;
    const double xd = x, yd = y, zd = z;
    // row . (x, y, z, 1), accumulated left to right in float64 like the 4 x 4 by 4 x N dot it restates
    x = (float)(((T[0] * xd + T[1] * yd) + T[2] * zd) + T[3]);
    y = (float)(((T[4] * xd + T[5] * yd) + T[6] * zd) + T[7]);
    z = (float)(((T[8] * xd + T[9] * yd) + T[10] * zd) + T[11]);
  }
  o[0] = x; o[1] = y; o[2] = z;
  for (int c = 3; c < use_dim; ++c) o[c] = c == tanh_column ? tanhf(p[c]) : p[c];
  o[use_dim] = time_lag[s];
}

extern "C" size_t pn_merge_sweeps_workspace_bytes(int n_points) {
  return pn_align_up((size_t)pn_div_up(n_points > 0 ? n_points : 1, SW_T) * 4, 256);
}

extern "C" int pn_merge_sweeps(const float* raw, int raw_dim, int use_dim, const int32_t* sweep_start_host,
                               int n_sweeps, const double* transform, const uint8_t* has_transform,
                               const float* time_lag, const uint8_t* filter_close, float radius, int tanh_column,
                               float* out, int32_t* out_count, void* workspace, size_t workspace_bytes, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(sweep_start_host && out_count && n_sweeps >= 1 && n_sweeps <= SW_MAX,
               "pn_merge_sweeps: 1..%d sweeps", SW_MAX);
  PN_CHECK_ARG(raw_dim >= 3 && use_dim >= 3 && use_dim <= raw_dim, "pn_merge_sweeps: bad point layout");
  PN_CHECK_ARG(tanh_column < 0 || (tanh_column >= 3 && tanh_column < use_dim),
               "pn_merge_sweeps: tanh_column %d is not a feature column (3..%d) or -1", tanh_column, use_dim - 1);
  SweepMeta M;
  M.n_sweeps = n_sweeps;
  for (int s = 0; s <= n_sweeps; ++s) {
    M.start[s] = sweep_start_host[s];
    PN_CHECK_ARG(s == 0 ? M.start[0] == 0 : M.start[s] >= M.start[s - 1], "pn_merge_sweeps: sweep offsets must ascend from 0");
  }
  const int total = M.start[n_sweeps];
  if (total == 0) {
    { const int rc_ = pn_fill_async(out_count, 0, 4, stream); if (rc_ != PN_OK) return rc_; }
    return PN_OK;
  }
  PN_CHECK_ARG(raw && transform && has_transform && time_lag && filter_close && out && workspace,
               "pn_merge_sweeps: null argument");
  if (workspace_bytes < pn_merge_sweeps_workspace_bytes(total)) {
    pn_set_error("pn_merge_sweeps: workspace too small");
    return PN_ENOSPC;
  }
  int32_t* block_sums = (int32_t*)workspace;
  const int nblocks = pn_div_up(total, SW_T);
  hipLaunchKernelGGL(sweeps_count_kernel, dim3(nblocks), dim3(SW_T), 0, stream, M, raw, raw_dim, filter_close,
                     radius, block_sums);
  PN_CHECK_LAUNCH("sweeps_count_kernel");
  hipLaunchKernelGGL(sweeps_scan_kernel, dim3(1), dim3(1024), 0, stream, block_sums, nblocks, out_count);
  PN_CHECK_LAUNCH("sweeps_scan_kernel");
  hipLaunchKernelGGL(sweeps_write_kernel, dim3(nblocks), dim3(SW_T), 0, stream, M, raw, raw_dim, use_dim, transform,
                     has_transform, time_lag, filter_close, radius, block_sums, tanh_column, out);
  PN_CHECK_LAUNCH("sweeps_write_kernel");
  return PN_OK;
}
