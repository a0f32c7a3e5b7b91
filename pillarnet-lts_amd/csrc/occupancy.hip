// BEV occupancy index: bit grid -> (word ranks, sorted (b,y,x) rows, count).
//
// This is what stands in for the reference's dense bool mask + int32 cumsum +
// `.item()` + position map (det3d/ops/pillar_ops/pillar_utils.py:38-48,
// createPillarIndicesKernel src/pillar_ops_gpu.cu:60-78): 1 bit per cell
// (259 KB for a 1440x1440 sample instead of 2 MB + 2 x 8.3 MB), two launches,
// the count never leaves the device.
#include "common.h"

static constexpr int SCAN_THREADS = 256;
static constexpr int SCAN_MAX_BLOCKS = 2048;

pnScanGeom pn_scan_geom(int64_t cells) {
  pnScanGeom g;
  g.words = (int)((cells + 63) / 64);
  g.words_per_thread = (int)((g.words + (int64_t)SCAN_THREADS * SCAN_MAX_BLOCKS - 1) /
                             ((int64_t)SCAN_THREADS * SCAN_MAX_BLOCKS));
  if (g.words_per_thread < 1) g.words_per_thread = 1;
  g.blocks = pn_div_up(g.words, (int64_t)SCAN_THREADS * g.words_per_thread);
  return g;
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < PN_WAVE; d <<= 1) {
    int t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive scan of one int per thread over a 256-thread block; returns the
// thread's exclusive prefix, *total = block sum (same in every thread).
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* lds /*>=8 ints*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int incl = wave_incl_scan(v, lane);
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  int wave_base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / PN_WAVE; ++w) {
    const int s = lds[w];
    if (w < wave) wave_base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return wave_base + incl - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void occ_block_sums_kernel(
    const unsigned long long* __restrict__ bits, int words, int wpt,
    int32_t* __restrict__ block_sums) {
  __shared__ int lds[8];
  const int w0 = (blockIdx.x * SCAN_THREADS + threadIdx.x) * wpt;
  int c = 0;
  for (int k = 0; k < wpt; ++k)
    if (w0 + k < words) c += __popcll(bits[w0 + k]);
  int total;
  block_excl_scan(c, &total, lds);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(SCAN_THREADS) void occ_rank_emit_kernel(
    const unsigned long long* __restrict__ bits, int words, int wpt, int H, int W,
    const int32_t* __restrict__ block_sums, int32_t* __restrict__ word_rank,
    int32_t* __restrict__ indices, int32_t* __restrict__ count,
    uint4* __restrict__ zero_rows, int zero_row_vec16, long long row_cap) {
  __shared__ int lds[8];
  // rows owned by earlier blocks
  int part = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_THREADS) part += block_sums[b];
  int base;
  block_excl_scan(part, &base, lds);

  const int w0 = (blockIdx.x * SCAN_THREADS + threadIdx.x) * wpt;
  int c = 0;
  for (int k = 0; k < wpt; ++k)
    if (w0 + k < words) c += __popcll(bits[w0 + k]);
  int total;
  int r = base + block_excl_scan(c, &total, lds);

  const int hw = H * W;
  for (int k = 0; k < wpt; ++k) {
    const int w = w0 + k;
    if (w >= words) break;
    unsigned long long m = bits[w];
    word_rank[w] = r;
    while (m) {
      const int bit = __builtin_ctzll(m);
      m &= m - 1ull;
      const int cell = w * 64 + bit;
      const int b = cell / hw;
      const int rem = cell - b * hw;
      const int y = rem / W;
      if (indices && r < row_cap) {
        indices[r * 3 + 0] = b;
        indices[r * 3 + 1] = y;
        indices[r * 3 + 2] = rem - y * W;
      }
      ++r;
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *count = base + total;

  // zero-fill this block's contiguous feature rows (scatter-max identity,
  // scatter_utils.py:20 `src.new_zeros(M, C)`) with coalesced 16-B stores
  if (zero_rows) {
    long long lo = (long long)base, hi = (long long)base + total;
    if (hi > row_cap) hi = row_cap;
    const long long v0 = lo * zero_row_vec16, v1 = hi * zero_row_vec16;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (long long v = v0 + threadIdx.x; v < v1; v += SCAN_THREADS) zero_rows[v] = z;
  }
}

int pn_occupancy_scan(const unsigned long long* bits, int64_t cells, int H, int W,
                      int32_t* word_rank, int32_t* block_sums, int32_t* indices,
                      int32_t* count, void* zero_rows, int zero_row_bytes, int64_t row_cap,
                      hipStream_t stream) {
  PN_CHECK_ARG(cells > 0 && cells < (1ll << 31), "occupancy: %lld cells out of range",
               (long long)cells);
  PN_CHECK_ARG(zero_row_bytes % 16 == 0, "occupancy: row bytes %d not a multiple of 16",
               zero_row_bytes);
  const pnScanGeom g = pn_scan_geom(cells);
  hipLaunchKernelGGL(occ_block_sums_kernel, dim3(g.blocks), dim3(SCAN_THREADS), 0, stream, bits,
                     g.words, g.words_per_thread, block_sums);
  PN_CHECK_LAUNCH("occ_block_sums_kernel");
  hipLaunchKernelGGL(occ_rank_emit_kernel, dim3(g.blocks), dim3(SCAN_THREADS), 0, stream, bits,
                     g.words, g.words_per_thread, H, W, block_sums, word_rank, indices, count,
                     (uint4*)zero_rows, zero_row_bytes / 16, (long long)row_cap);
  PN_CHECK_LAUNCH("occ_rank_emit_kernel");
  return PN_OK;
}

// occupancy index of an existing (b,y,x) row list (rows must already be in ascending flat
// order, which is what every producer in this library emits)
__global__ __launch_bounds__(SCAN_THREADS) void occ_mark_rows_kernel(
    const int32_t* __restrict__ indices, const int32_t* __restrict__ count, int cap, int B, int H,
    int W, unsigned long long* __restrict__ bits) {
  int M = *count;
  if (M > cap) M = cap;
  for (int m = blockIdx.x * SCAN_THREADS + threadIdx.x; m < M; m += gridDim.x * SCAN_THREADS) {
    const int b = indices[m * 3 + 0], y = indices[m * 3 + 1], x = indices[m * 3 + 2];
    if (b < 0 || b >= B || y < 0 || y >= H || x < 0 || x >= W) continue;
    const int cell = (b * H + y) * W + x;
    atomicOr(&bits[cell >> 6], 1ull << (cell & 63));
  }
}

extern "C" size_t pn_occupancy_scratch_bytes(int batch, int H, int W) {
  const pnScanGeom sg = pn_scan_geom((int64_t)batch * H * W);
  return pn_align_up((size_t)sg.blocks * 4, 256);
}

extern "C" int pn_occupancy_from_indices(const int32_t* indices, const int32_t* count_dev, int cap,
                                         int batch, int H, int W, uint64_t* occ_bits,
                                         int32_t* occ_rank, int32_t* active_cells_dev,
                                         void* scratch, size_t scratch_bytes, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(indices && count_dev && occ_bits && occ_rank && active_cells_dev && scratch,
               "pn_occupancy_from_indices: null argument");
  const int64_t cells = (int64_t)batch * H * W;
  PN_CHECK_ARG(cap >= 1 && cells > 0 && cells < (1ll << 31), "pn_occupancy_from_indices: bad shape");
  if (scratch_bytes < pn_occupancy_scratch_bytes(batch, H, W)) {
    pn_set_error("pn_occupancy_from_indices: scratch too small");
    return PN_ENOSPC;
  }
  const pnScanGeom sg = pn_scan_geom(cells);
  PN_CHECK_HIP(hipMemsetAsync(occ_bits, 0, (size_t)sg.words * 8, stream));
  int blocks = pn_div_up(cap, SCAN_THREADS);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(occ_mark_rows_kernel, dim3(blocks), dim3(SCAN_THREADS), 0, stream, indices,
                     count_dev, cap, batch, H, W, (unsigned long long*)occ_bits);
  PN_CHECK_LAUNCH("occ_mark_rows_kernel");
  return pn_occupancy_scan((const unsigned long long*)occ_bits, cells, H, W, occ_rank,
                           (int32_t*)scratch, nullptr, active_cells_dev, nullptr, 0, cap, stream);
}
