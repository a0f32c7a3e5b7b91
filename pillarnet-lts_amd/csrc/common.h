// Shared host/device helpers for the gfx950 pillar-detection kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "pillarnet_hip.h"

#define PN_WAVE 64

void pn_set_error(const char* fmt, ...);

#define PN_CHECK_ARG(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      pn_set_error(__VA_ARGS__);         \
      return PN_EINVAL;                  \
    }                                    \
  } while (0)

#define PN_CHECK_HIP(expr)                                                   \
  do {                                                                       \
    hipError_t e__ = (expr);                                                 \
    if (e__ != hipSuccess) {                                                 \
      pn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),  \
                   __FILE__, __LINE__);                                      \
      return (int)e__;                                                       \
    }                                                                        \
  } while (0)

#define PN_CHECK_LAUNCH(name)                                                \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      pn_set_error("launch of %s failed: %s", name, hipGetErrorString(e__)); \
      return (int)e__;                                                       \
    }                                                                        \
  } while (0)

// capi.hip: memset as a kernel (safe to capture into a HIP graph)
int pn_fill_async(void* ptr, int byte, size_t bytes, hipStream_t stream);

static inline size_t pn_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int pn_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// A BEV occupancy index: one bit per cell plus, per 64-cell word, the number of
// active cells in all earlier words.  rank(cell) = word_rank[cell>>6] +
// popcount(bits[cell>>6] below the cell's bit) is the cell's row in the
// ascending-flat-index pillar list (the order pillar_utils.py:43-48 produces
// with cumsum), so no dense int32 [B,H,W] position map is ever built.
struct pnOccupancy {
  const unsigned long long* bits;
  const int32_t* word_rank;
};

__device__ __forceinline__ int pn_occ_rank(const unsigned long long* __restrict__ bits,
                                           const int32_t* __restrict__ word_rank, int cell) {
  const int w = cell >> 6;
  const unsigned long long m = bits[w];
  const unsigned long long below = m & ((1ull << (cell & 63)) - 1ull);
  return ((m >> (cell & 63)) & 1ull) ? word_rank[w] + __popcll(below) : -1;
}

// Launch-size helper shared by the scan kernels (see occupancy.hip).
struct pnScanGeom {
  int words;            // 64-cell words
  int words_per_thread; // contiguous words one thread owns
  int blocks;           // 256-thread blocks
};
pnScanGeom pn_scan_geom(int64_t cells);

// occupancy.hip: (bits | byte map | finer level's bits) -> bits, word_rank, (b,y,x) rows, count;
// optionally zero-fills feature rows [0, count) with `zero_row_bytes` bytes each.
int pn_occupancy_from_bytes(const unsigned char* bytes, unsigned long long* bits, int64_t cells, int H, int W,
                            int32_t* word_rank, int32_t* block_sums, int32_t* indices, int32_t* count,
                            void* zero_rows, int zero_row_bytes, int64_t row_cap, hipStream_t stream);
int pn_occupancy_downsample(const unsigned long long* fine_bits, int Hi, int Wi, int stride, int pad,
                            unsigned long long* bits, int batch, int Ho, int Wo, int32_t* word_rank,
                            int32_t* block_sums, int32_t* indices, int32_t* count, int64_t row_cap,
                            hipStream_t stream);
int pn_occupancy_scan(const unsigned long long* bits, int64_t cells, int H, int W,
                      int32_t* word_rank, int32_t* block_sums, int32_t* indices,
                      int32_t* count, void* zero_rows, int zero_row_bytes, int64_t row_cap,
                      hipStream_t stream);
