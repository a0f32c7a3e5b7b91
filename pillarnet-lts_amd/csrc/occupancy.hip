// BEV occupancy index: one bit per cell + per-word rank -> sorted (b,y,x) rows and their count.
//
// This is what stands in for the reference's dense bool mask + int32 cumsum + `.item()` + position
// map (det3d/ops/pillar_ops/pillar_utils.py:38-48, createPillarIndicesKernel
// src/pillar_ops_gpu.cu:60-78) and for spconv's hash-table active-set construction of a strided
// conv.  Two launches per level, one thread per CELL, no atomics, the count never leaves the device:
//
//   occ_words_kernel   lane = cell, wavefront = one 64-cell word: the cell's activity comes from
//                      (a) a byte map the point pass filled with plain stores, (b) the 3x3 / stride-s
//                      window of a finer level's bits (the output active set of SparseConv2d), or
//                      (c) existing bits; `__ballot` IS the word.  Block popcounts -> block_sums.
//   occ_emit_kernel    block base = sum of earlier block_sums; word rank = base + scan of the 16
//                      words of the block; an active lane's row = rank + popcount(lower lanes), so
//                      (b,y,x) rows come out sorted by flat index with coalesced stores.  Optionally
//                      zero-fills the block's feature rows (scatter-max identity).
#include "common.h"

static constexpr int OCC_THREADS = 1024;                 // 16 words per block
static constexpr int OCC_WORDS = OCC_THREADS / 64;

pnScanGeom pn_scan_geom(int64_t cells) {
  pnScanGeom g;
  g.words = (int)((cells + 63) / 64);
  g.words_per_thread = 1;
  g.blocks = pn_div_up(cells, OCC_THREADS);
  return g;
}

enum { OCC_FROM_BITS = 0, OCC_FROM_BYTES = 1, OCC_FROM_FINER = 2 };

struct OccSource {
  int kind;
  const unsigned char* bytes;            // OCC_FROM_BYTES: [cells]
  const unsigned long long* fine_bits;   // OCC_FROM_FINER: bits of the input level
  int Hi, Wi, stride, pad, Ho, Wo;
};

__device__ __forceinline__ bool fine_bit(const unsigned long long* __restrict__ bits, int cell) {
  return (bits[cell >> 6] >> (cell & 63)) & 1ull;
}

__global__ __launch_bounds__(OCC_THREADS) void occ_words_kernel(OccSource src, long long cells,
                                                                unsigned long long* __restrict__ bits,
                                                                int32_t* __restrict__ block_sums) {
  __shared__ int wsum[OCC_WORDS];
  const long long cell = (long long)blockIdx.x * OCC_THREADS + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long w = cell >> 6;
  unsigned long long word;
  if (src.kind == OCC_FROM_BITS) {
    word = cell - lane < cells ? bits[w] : 0ull;
  } else {
    bool active = false;
    if (cell < cells) {
      if (src.kind == OCC_FROM_BYTES) {
        active = src.bytes[cell] != 0;
      } else {
        const int hw = src.Ho * src.Wo;
        const int b = (int)(cell / hw), rem = (int)(cell - (long long)b * hw);
        const int oy = rem / src.Wo, ox = rem - oy * src.Wo;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = oy * src.stride - src.pad + ky;
          if (iy < 0 || iy >= src.Hi) continue;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * src.stride - src.pad + kx;
            if (ix < 0 || ix >= src.Wi) continue;
            active |= fine_bit(src.fine_bits, (b * src.Hi + iy) * src.Wi + ix);
          }
        }
      }
    }
    word = __ballot(active);
    if (lane == 0 && cell < cells) bits[w] = word;
  }
  if (lane == 0) wsum[wave] = __popcll(word);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int i = 0; i < OCC_WORDS; ++i) t += wsum[i];
    block_sums[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(OCC_THREADS) void occ_emit_kernel(
    const unsigned long long* __restrict__ bits, long long cells, int H, int W,
    const int32_t* __restrict__ block_sums, int32_t* __restrict__ word_rank,
    int32_t* __restrict__ indices, int32_t* __restrict__ count, uint4* __restrict__ zero_rows,
    int zero_row_vec16, long long row_cap) {
  __shared__ int red[OCC_WORDS];
  __shared__ int wpre[OCC_WORDS + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // rows owned by earlier blocks
  int part = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += OCC_THREADS) part += block_sums[b];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
  if (lane == 0) red[wave] = part;
  const long long cell = (long long)blockIdx.x * OCC_THREADS + threadIdx.x;
  const long long w = cell >> 6;
  const unsigned long long word = cell - lane < cells ? bits[w] : 0ull;
  __syncthreads();
  if (threadIdx.x == 0) {
    int base = 0;
#pragma unroll
    for (int i = 0; i < OCC_WORDS; ++i) base += red[i];
    wpre[0] = base;
  }
  __syncthreads();
  if (lane == 0) red[wave] = __popcll(word);
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = wpre[0];
#pragma unroll
    for (int i = 0; i < OCC_WORDS; ++i) {
      const int c = red[i];
      wpre[i] = run;
      run += c;
    }
    wpre[OCC_WORDS] = run;
  }
  __syncthreads();
  const int wr = wpre[wave];
  if (lane == 0 && cell < cells) word_rank[w] = wr;
  if (indices && ((word >> lane) & 1ull)) {
    const long long r = wr + __popcll(word & ((1ull << lane) - 1ull));
    if (r < row_cap) {
      const int hw = H * W;
      const int b = (int)(cell / hw), rem = (int)(cell - (long long)b * hw);
      const int y = rem / W;
      indices[r * 3 + 0] = b;
      indices[r * 3 + 1] = y;
      indices[r * 3 + 2] = rem - y * W;
    }
  }
  const int base = wpre[0], total = wpre[OCC_WORDS] - wpre[0];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *count = base + total;
  // zero-fill this block's contiguous feature rows (scatter_utils.py:20 `src.new_zeros(M, C)`)
  if (zero_rows) {
    long long lo = (long long)base, hi = (long long)base + total;
    if (hi > row_cap) hi = row_cap;
    const long long v0 = lo * zero_row_vec16, v1 = hi * zero_row_vec16;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (long long v = v0 + threadIdx.x; v < v1; v += OCC_THREADS) zero_rows[v] = z;
  }
}

static int occ_build(const OccSource& src, unsigned long long* bits, int64_t cells, int H, int W,
                     int32_t* word_rank, int32_t* block_sums, int32_t* indices, int32_t* count,
                     void* zero_rows, int zero_row_bytes, int64_t row_cap, hipStream_t stream) {
  PN_CHECK_ARG(cells > 0 && cells < (1ll << 31), "occupancy: %lld cells out of range", (long long)cells);
  PN_CHECK_ARG(zero_row_bytes % 16 == 0, "occupancy: row bytes %d not a multiple of 16", zero_row_bytes);
  const pnScanGeom g = pn_scan_geom(cells);
  hipLaunchKernelGGL(occ_words_kernel, dim3(g.blocks), dim3(OCC_THREADS), 0, stream, src, (long long)cells, bits,
                     block_sums);
  PN_CHECK_LAUNCH("occ_words_kernel");
  hipLaunchKernelGGL(occ_emit_kernel, dim3(g.blocks), dim3(OCC_THREADS), 0, stream, bits, (long long)cells, H, W,
                     block_sums, word_rank, indices, count, (uint4*)zero_rows, zero_row_bytes / 16,
                     (long long)row_cap);
  PN_CHECK_LAUNCH("occ_emit_kernel");
  return PN_OK;
}

int pn_occupancy_scan(const unsigned long long* bits, int64_t cells, int H, int W, int32_t* word_rank,
                      int32_t* block_sums, int32_t* indices, int32_t* count, void* zero_rows,
                      int zero_row_bytes, int64_t row_cap, hipStream_t stream) {
  OccSource s = {};
  s.kind = OCC_FROM_BITS;
  return occ_build(s, const_cast<unsigned long long*>(bits), cells, H, W, word_rank, block_sums, indices, count,
                   zero_rows, zero_row_bytes, row_cap, stream);
}

int pn_occupancy_from_bytes(const unsigned char* bytes, unsigned long long* bits, int64_t cells, int H, int W,
                            int32_t* word_rank, int32_t* block_sums, int32_t* indices, int32_t* count,
                            void* zero_rows, int zero_row_bytes, int64_t row_cap, hipStream_t stream) {
  OccSource s = {};
  s.kind = OCC_FROM_BYTES;
  s.bytes = bytes;
  return occ_build(s, bits, cells, H, W, word_rank, block_sums, indices, count, zero_rows, zero_row_bytes,
                   row_cap, stream);
}

int pn_occupancy_downsample(const unsigned long long* fine_bits, int Hi, int Wi, int stride, int pad,
                            unsigned long long* bits, int batch, int Ho, int Wo, int32_t* word_rank,
                            int32_t* block_sums, int32_t* indices, int32_t* count, int64_t row_cap,
                            hipStream_t stream) {
  OccSource s = {};
  s.kind = OCC_FROM_FINER;
  s.fine_bits = fine_bits;
  s.Hi = Hi; s.Wi = Wi; s.stride = stride; s.pad = pad; s.Ho = Ho; s.Wo = Wo;
  return occ_build(s, bits, (int64_t)batch * Ho * Wo, Ho, Wo, word_rank, block_sums, indices, count, nullptr, 0,
                   row_cap, stream);
}

// occupancy index of an existing (b,y,x) row list (rows must already be in ascending flat
// order, which is what every producer in this library emits)
__global__ __launch_bounds__(256) void occ_mark_rows_kernel(const int32_t* __restrict__ indices,
                                                            const int32_t* __restrict__ count, int cap, int B,
                                                            int H, int W, unsigned long long* __restrict__ bits) {
  int M = *count;
  if (M > cap) M = cap;
  for (int m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
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
  { const int rc_ = pn_fill_async(occ_bits, 0, (size_t)sg.words * 8, stream); if (rc_ != PN_OK) return rc_; }
  int blocks = pn_div_up(cap, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(occ_mark_rows_kernel, dim3(blocks), dim3(256), 0, stream, indices, count_dev, cap, batch, H,
                     W, (unsigned long long*)occ_bits);
  PN_CHECK_LAUNCH("occ_mark_rows_kernel");
  return pn_occupancy_scan((const unsigned long long*)occ_bits, cells, H, W, occ_rank, (int32_t*)scratch, nullptr,
                           active_cells_dev, nullptr, 0, cap, stream);
}
