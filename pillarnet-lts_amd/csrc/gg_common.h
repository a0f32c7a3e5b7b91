// Types shared by the two gather-GEMM kernels (gather_gemm.hip: per-tap gather staging;
// gather_gemm_win.hip: contiguous-window staging).
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int BK> __device__ __forceinline__ int swz(int row);
template <> __device__ __forceinline__ int swz<64>(int row) { return (row >> 1) & 7; }  // 8 chunks / row
template <> __device__ __forceinline__ int swz<32>(int row) { return (row >> 2) & 3; }  // 4 chunks / row

template <typename TOut> struct OutOps;
template <> struct OutOps<__bf16> {
  static __device__ __forceinline__ void store4(__bf16* p, f32x4 v) {
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
    *reinterpret_cast<bf16x4*>(p) = o;
  }
};
template <> struct OutOps<float> {
  static __device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};

struct GGArgs {
  const __bf16* in;        // [*, in_ld] rows, the first cin channels are used
  int in_ld;
  unsigned in_bytes;       // extent of `in` for the buffer descriptor of the LDS-DMA path (0: use register staging)
  const int32_t* nbr;      // [ntaps][nbr_stride] input row per (tap, output site) or -1; NULL: identity (ntaps == 1)
  int nbr_stride;
  const int32_t* count;    // device row count (NULL: rows)
  int rows;                // capacity / host row count
  const __bf16* wt;        // [ntaps][cout_total][cin]
  int cin, cout_total, ntaps;
  const float* scale;      // [cout_total]
  const float* shift;
  const __bf16* residual;  // [rows, res_ld] or NULL
  int res_ld;
  int relu;
  void* out;               // [*, out_ld] rows, channel offset already applied by the caller
  int out_ld;
  const int32_t* out_rows; // optional scatter: output row of site m (NULL: m)
  // grouped launch (blockIdx.z = group g): element offsets added per group
  long long g_in, g_wt, g_ss, g_out, g_rows, g_nbr, g_count;
  int diag;                // 1: skip LDS reads + MFMAs (staging-rate probe), 2: skip staging (math-rate probe)
  int free_order;          // 1: the caller compares this conv with no other launch shape of the same weights (a sparse
                           // backbone level), so the tile may pick its own channel chunking; 0: the summation order is
                           // a function of (cin, taps) alone
};

// set by pn_spconv_forward around its pn_gather_gemm call (host threads launch independently)
extern thread_local int pn_gg_free_order;

// out-of-range offset: a buffer load through a descriptor drops it and delivers zeros -- how a
// missing neighbour (or a weight row past cout) becomes a zero LDS row on the LDS-DMA path
static constexpr unsigned GG_OOB = 0x80000000u;
static constexpr int GG_MAX_TAPS = 9;


// gather_gemm_win.hip: returns PN_OK after launching, or -1 when the problem is not eligible
// (caller falls back to the per-tap kernel).
int pn_gg_try_window(const GGArgs& a, int groups, bool out_fp32, hipStream_t stream);
// gather_gemm_direct.hip: the 32 -> 32 level without LDS staging; same return convention
int pn_gg_try_direct(const GGArgs& a, int groups, bool out_fp32, hipStream_t stream);
