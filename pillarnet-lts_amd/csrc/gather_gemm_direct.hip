// Direct-gather implicit GEMM for the 32 -> 32 (and 32 -> 64) rulebook convolutions of the first backbone level (bf16 in / out,
// fp32 accumulate), gfx950.  Same contract as gather_gemm.hip.
//
// The first level is the sparsest of the frame: ~2 * 10^5 pillars on a 1440^2 grid, three of a site's nine taps
// present on average.  Its MFMA work is nothing (6 GFLOP issued); what a launch costs is latency.  The window
// kernel pays two dependent global round trips before a tile's first MFMA (rulebook tile -> run bounds -> window
// DMA), three workgroup barriers per unit and a parked epilogue, for 12 MFMAs per wave and unit: 18 us against
// an HBM floor of ~6.  Here nothing is staged and nothing is shared but the weights:
//   * the nine 32 x 32 tap matrices (18 KB) sit in LDS for the whole workgroup, one barrier after loading them;
//   * a wave owns 32 consecutive output sites; lane l holds site l % 32, reads its nine rulebook entries, and
//     fetches its MFMA B operand -- 16 bytes of the neighbour's 64-byte row -- straight from global memory with a
//     buffer load (a missing neighbour is an out-of-range offset: the load returns zeros, no branch); all 18 loads
//     are in flight before the first MFMA, behind them only the rulebook read: two dependent round trips per wave,
//     with the weight, scale / shift and residual reads issued beside the first;
//   * the accumulators go through a wave-private 4.5 KB LDS tile (no workgroup barrier) so that four lanes write a
//     whole 64-byte output row and read the residual row the same way.
// Summation order = tap-major, two 16-channel k-steps per tap: the order of the window kernel's unit stages, so the
// two kernels agree bit for bit.
#include "gg_common.h"
#include <stdlib.h>

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#endif

static constexpr int GD_PARK_LD = 36;       // floats per parked site row (32 + 4: the rows of a lane group on different banks)

// NB = output channels / 32: the blocks are computed one after the other from the same operand registers.
// GD_WAVES = waves per workgroup (a wave owns 32 consecutive output sites): 4 for 32 outputs (11.9 against 13.2 us
// with 8 on the 2 * 10^5-row level: more workgroups per CU), 8 for 64 (18.9 against 21.9: the 36 KB of weights
// are fetched half as often).
template <int NB, int GD_WAVES>
__global__ __launch_bounds__(GD_WAVES * 64, 4) void gg_direct32_kernel(GGArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char wsm[9 * 32 * NB * 64];                  // [tap][cout][cin] bf16, chunks swizzled
  __shared__ __attribute__((aligned(16))) float park[GD_WAVES][32 * GD_PARK_LD];
  __shared__ __attribute__((aligned(16))) float ss_lds[2 * 32 * NB];                   // scale, shift
  constexpr int NT = GD_WAVES * 64, WROWS = 9 * 32 * NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  int M = a.count ? *a.count : a.rows;
  if (M > a.rows) M = a.rows;
  if ((int)blockIdx.x * GD_WAVES * 32 >= M) return;           // workgroup-uniform
  const int mb = (blockIdx.x * GD_WAVES + wave) * 32;          // this wave's sites (may lie past M: it then only helps load)
  const int site = mb + lr;
  // first round trip: rulebook entries and the weights
  int src[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) src[t] = site < M ? a.nbr[(size_t)t * a.nbr_stride + site] : -1;
  constexpr int WCH = (WROWS * 4 + NT - 1) / NT;               // 16-B weight chunks per thread
  if (tid < 32 * NB) {
    ss_lds[tid] = a.scale[tid];
    ss_lds[32 * NB + tid] = a.shift[tid];
  }
  {
    bf16x8 wv[WCH];
#pragma unroll
    for (int j = 0; j < WCH; ++j) {
      const int q = tid + j * NT;
      if (q < WROWS * 4) wv[j] = *reinterpret_cast<const bf16x8*>(a.wt + (size_t)(q >> 2) * 32 + (q & 3) * 8);
    }
#pragma unroll
    for (int j = 0; j < WCH; ++j) {
      const int q = tid + j * NT, row = q >> 2, ch = q & 3;
      if (q < WROWS * 4) *reinterpret_cast<bf16x8*>(wsm + row * 64 + ((ch ^ swz<32>(row & 31)) << 4)) = wv[j];
    }
  }
  // second round trip: the gathered operand rows -- 16 bytes of each neighbour's 64-byte row per lane, zeros for a
  // missing neighbour
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
  const unsigned in_row_bytes = (unsigned)a.in_ld * 2u;
  u32x4 x[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#if defined(GD_DIAG) && GD_DIAG == 2     // tuning probe: every lane of a tap reads row `lr` (no divergence between rows)
      const unsigned off = (unsigned)lr * in_row_bytes + (unsigned)(32 * ks + 16 * lh) + (src[t] >= 0 ? 0u : 0u);
#else
      const unsigned off = src[t] >= 0 ? (unsigned)src[t] * in_row_bytes + (unsigned)(32 * ks + 16 * lh) : GG_OOB;
#endif
#if defined(GD_DIAG) && GD_DIAG == 1     // tuning probe: no operand loads at all
      x[t][ks] = u32x4{(unsigned)src[t], off, 0u, 0u};
#else
      x[t][ks] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
#endif
    }
  // this lane's epilogue role: site lane / 4 (+16 in the second pass), channels 8 * (lane % 4) .. +8 of a 32-channel block
  const int e_site = lane >> 2, e_ch = (lane & 3) * 8;
  __syncthreads();
  if (mb >= M) return;                                         // wave-uniform, after the only barrier
  float* const pk = park[wave];
  __bf16* const out = reinterpret_cast<__bf16*>(a.out);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int c0 = 32 * nb + e_ch;
    bf16x8 rv[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int m = mb + e_site + 16 * p;
#pragma unroll
      for (int r = 0; r < 8; ++r) rv[p][r] = (__bf16)0.f;
      if (a.residual && m < M) rv[p] = *reinterpret_cast<const bf16x8*>(a.residual + (size_t)m * a.res_ld + c0);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int row = (t * NB + nb) * 32 + lr;               // weight layout [tap][cout][cin]
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(wsm + row * 64 + (((2 * ks + lh) ^ swz<32>(lr)) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, __builtin_bit_cast(bf16x8, x[t][ks]), acc, 0, 0, 0);
      }
    // D row = output channel (r & 3) + 8 * (r >> 2) + 4 * lh, D column = site lr -> parked [site][channel] fp32
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[4 * g + r];
      *reinterpret_cast<f32x4*>(pk + lr * GD_PARK_LD + 8 * g + 4 * lh) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // wave-private tile: the wave's own writes have landed
    f32x4 sc[2], sh[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      sc[h] = *reinterpret_cast<const f32x4*>(ss_lds + c0 + 4 * h);
      sh[h] = *reinterpret_cast<const f32x4*>(ss_lds + 32 * NB + c0 + 4 * h);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int sl = e_site + 16 * p, m = mb + sl;
      f32x4 v0 = *reinterpret_cast<const f32x4*>(pk + sl * GD_PARK_LD + e_ch) * sc[0] + sh[0];
      f32x4 v1 = *reinterpret_cast<const f32x4*>(pk + sl * GD_PARK_LD + e_ch + 4) * sc[1] + sh[1];
      if (m < M) {
        if (a.residual) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { v0[r] += (float)rv[p][r]; v1[r] += (float)rv[p][4 + r]; }
        }
        if (a.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { v0[r] = fmaxf(v0[r], 0.f); v1[r] = fmaxf(v1[r], 0.f); }
        }
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) { o[r] = (__bf16)v0[r]; o[4 + r] = (__bf16)v1[r]; }
        *reinterpret_cast<bf16x8*>(out + (size_t)m * a.out_ld + c0) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the tile is read before the next block overwrites it
  }
#endif
}

// PN_OK after launching, -1 when the problem is not this kernel's (the caller goes on to the window kernel)
int pn_gg_try_direct(const GGArgs& a, int groups, bool out_fp32, hipStream_t stream) {
  static const bool off = getenv("PN_GG_NO_DIRECT") != nullptr;
  if (off || out_fp32 || groups != 1 || a.ntaps != 9 || a.nbr == nullptr || a.cin != 32) return -1;
  if (a.cout_total != 32 && a.cout_total != 64) return -1;
  if (a.in_bytes == 0 || a.out_rows != nullptr || a.diag != 0) return -1;
  if ((a.in_ld & 7) != 0 || (a.out_ld & 7) != 0 || (a.residual && (a.res_ld & 7) != 0)) return -1;
  if ((((uintptr_t)a.out | (uintptr_t)a.residual | (uintptr_t)a.scale | (uintptr_t)a.shift | (uintptr_t)a.wt) & 15) != 0) return -1;
  if (a.cout_total == 32)
    hipLaunchKernelGGL((gg_direct32_kernel<1, 4>), dim3(pn_div_up(a.rows, 4 * 32)), dim3(4 * 64), 0, stream, a);
  else
    hipLaunchKernelGGL((gg_direct32_kernel<2, 8>), dim3(pn_div_up(a.rows, 8 * 32)), dim3(8 * 64), 0, stream, a);
  PN_CHECK_LAUNCH("gg_direct32_kernel");
  return PN_OK;
}
