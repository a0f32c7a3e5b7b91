// LDS-tiled gather -> MFMA implicit GEMM (bf16 in, fp32 accumulate) for gfx950.
//
//   out[m, n0:n0+BN] = epilogue( sum_{tap < ntaps} W[tap][n, :] . in[nbr[tap][m], :] )
//
// Serves the sparse backbone (rulebook from the occupancy index) and -- with a regular-grid
// rulebook -- the dense 3x3 / transposed convs of conv5, the neck and the head, whose activations
// already live as [pixels, channels] rows (channels-last).
//
// Structure (one workgroup = BM output sites x BN output channels, NWM x NWN wavefronts):
//   * K is walked in stages of BK input channels of one tap.  Per stage the workgroup stages the
//     BN x BK weight tile and the BM x BK *gathered* activation tile into LDS (16-B global loads
//     into registers issued one stage ahead, written to the other LDS buffer after the MFMAs of
//     the current stage: one barrier per stage).  Missing neighbours contribute zero rows.
//   * LDS rows are BK bf16 (128 B for BK = 64).  The 16-B chunk index is XOR-swizzled with the
//     row so that the four 16-lane groups of a ds_read_b128 each touch all 64 banks once.
//   * MFMA v_mfma_f32_32x32x16_bf16 computes D^T: A = weight rows (32 output channels x 16 k),
//     B = activation rows (16 k x 32 sites).  A lane then owns 4 x 4 consecutive channels of one
//     site, so the epilogue (per-channel scale/shift = folded bias + eval BatchNorm, optional
//     residual, ReLU) stores 8-B (bf16) / 16-B (fp32) pieces straight into row-major output,
//     with a row stride and channel offset so "concat" never materialises.
//   * Weights are re-read from L2 once per workgroup, not once per wavefront as in the first
//     kernel (spconv.hip), which was what bounded it at ~7 % of the MFMA peak.
#include "gg_common.h"
#include <stdlib.h>

// diag == 3: wave 0 of the first workgroups stamps s_memtime at the phase boundaries of every stage
// (read back with pn_gg_debug_stamps; a tuning aid, not part of the public header)
static constexpr int GG_STAMP_WG = 4, GG_STAMP_STAGES = 40, GG_STAMP_PER = 6;
__device__ unsigned long long gg_stamps[GG_STAMP_WG * GG_STAMP_STAGES * GG_STAMP_PER];
// Probe builds only (-DGGW_STAMPS, scripts/build_probe_lib.sh): in the shipped kernels the
// stamps and the probes that drop the MFMAs (diag 1) or the staging (diag 2) are compiled out -- a scalar branch per
// hook costs the stage loop ~20 cycles.
#ifdef GGW_STAMPS
#define GG_DIAG(x) (a.diag == (x))
#define GG_STAMP(slot)                                                                                  \
  do {                                                                                                  \
    if (a.diag == 3 && wave == 0 && blockIdx.x < GG_STAMP_WG && s < GG_STAMP_STAGES && blockIdx.y == 0) { \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                       \
      if (lane == 0) gg_stamps[(blockIdx.x * GG_STAMP_STAGES + s) * GG_STAMP_PER + (slot)] = t_;         \
    }                                                                                                   \
  } while (0)
#else
#define GG_DIAG(x) false
#define GG_STAMP(slot) do {} while (0)
#endif

extern "C" int pn_gg_debug_stamps(unsigned long long* host_out, int n) {
  const int total = GG_STAMP_WG * GG_STAMP_STAGES * GG_STAMP_PER;
  if (n > total) n = total;
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gg_stamps), (size_t)n * 8);
}

template <int BM, int BN, int NWM, int NWN, int BK, typename TOut, bool DMA, int NBUF>
__global__ __launch_bounds__(NWM * NWN * 64) void gather_gemm_kernel(GGArgs a) {
  static_assert(NBUF == 2 || (NBUF == 3 && DMA), "3-deep ring only on the LDS-DMA path");
  constexpr int NT = NWM * NWN * 64;
  constexpr int CH = BK / 8;                 // 16-B chunks per LDS row
  constexpr int WTM = BM / NWM, WTN = BN / NWN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int ROWB = BK * 2;               // bytes per LDS row
  constexpr int TILE_CHUNKS = (BM + BN) * CH;
  constexpr int NLD = (TILE_CHUNKS + NT - 1) / NT;
  constexpr int BUF_BYTES = (BM + BN) * ROWB;
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be made of 32x32 MFMA tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* nbr_lds = reinterpret_cast<int*>(smem + NBUF * BUF_BYTES);   // [ntaps][BM] rulebook tile

  {
    const long long g = blockIdx.z;
    a.in += g * a.g_in;
    a.wt += g * a.g_wt;
    a.scale += g * a.g_ss;
    a.shift += g * a.g_ss;
    a.out = reinterpret_cast<char*>(a.out) + g * a.g_out * (long long)sizeof(TOut);
    if (a.out_rows) a.out_rows += g * a.g_rows;
    if (a.nbr) a.nbr += g * a.g_nbr;
    if (a.count) a.count += g * a.g_count;
  }
  int M = a.count ? *a.count : a.rows;
  if (M > a.rows) M = a.rows;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MiB
  // L2), so give every XCD one CONTIGUOUS band of row tiles.  Neighbouring tiles gather overlapping
  // input rows (3x3 halo: ~3 image rows per 0.7-row tile); in a band they hit the same L2 instead of
  // every L2 pulling the whole map.  Bijective for any grid size; affects speed only.
  int bx = blockIdx.x;
  {
    const int nwg = (M + BM - 1) / BM;   // tiles that hold rows (the grid covers the capacity)
    if (bx >= nwg) return;
    const int x = bx & 7, q = nwg >> 3, r = nwg & 7;
    bx = x * q + (x < r ? x : r) + (bx >> 3);
  }
  const int m0 = bx * BM;
  const int n0 = blockIdx.y * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  // wave index as a scalar: keeps the LDS-DMA descriptor choice, M0 base and branches wave-uniform
  // (otherwise hipcc wraps every buffer_load in a readfirstlane waterfall loop)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int lr = lane & 31, lh = lane >> 5;
  const int kchunks = a.cin / BK;            // stages per tap
  const int stages = a.ntaps * kchunks;

  // ---- this thread's share of a stage: which (row, chunk) pieces it moves -----------------------
  int ld_row[NLD], ld_chunk[NLD];
  bool ld_act[NLD], ld_ok[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int q = tid + j * NT;
    ld_ok[j] = q < TILE_CHUNKS;
    ld_act[j] = q < BM * CH;
    const int qq = ld_act[j] ? q : q - BM * CH;
    ld_row[j] = qq / CH;
    ld_chunk[j] = qq % CH;
  }
  uint4 regs[NLD];

  // the tile's rulebook goes to LDS once: the per-stage gather then costs one LDS read + one
  // global load instead of two dependent global loads
  for (int q = tid; q < a.ntaps * BM; q += NT) {
    const int tap = q / BM, r = q - tap * BM, m = m0 + r;
    int src = -1;
    if (m < M) src = a.nbr ? a.nbr[(size_t)tap * a.nbr_stride + m] : m;
    nbr_lds[q] = src;
  }
  __syncthreads();

  // ---- LDS-DMA staging (buffer_load_dwordx4 ... lds): each wave-instruction lands 1 KiB = 8 LDS rows;
  // lane l fills slot l%8 of row 8p + l/8, so it fetches chunk (slot ^ swizzle(row)) of that row's
  // source -- the swizzle lives in the per-lane SOURCE address, the LDS image stays linear.
  constexpr int PIECES = (BM + BN) / 8;            // per stage (rows are BK*2 = 8*16 B or 4*16 B)
  constexpr int ROWS_PER_PIECE = 1024 / ROWB;      // 8 (BK 64) or 16 (BK 32)
  constexpr int NPIECE = ((BM + BN) / ROWS_PER_PIECE + NT / 64 - 1) / (NT / 64);
  __amdgpu_buffer_rsrc_t rs_in, rs_wt;
  if constexpr (DMA) {
    rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    rs_wt = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, (int)((size_t)a.ntaps * a.cout_total * a.cin * 2), 0x00020000);
  }
  (void)PIECES;
  auto issue_dma = [&](int s, int buf) {
    const int tap = s / kchunks, kc = s - tap * kchunks;
    char* base = smem + buf * BUF_BYTES;
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      const int piece = wave + j * (NT / 64);       // wave-uniform
      if (piece * ROWS_PER_PIECE >= BM + BN) break;
      const int row = piece * ROWS_PER_PIECE + lane / CH;   // row inside [acts | weights]
      const int slot = lane % CH;
      auto lds = (__attribute__((address_space(3))) void*)(base + piece * 1024);
      if (piece * ROWS_PER_PIECE < BM) {   // wave-uniform: a piece never straddles the two tiles
        const int src = nbr_lds[tap * BM + row];
        const unsigned off = src >= 0 ? (unsigned)src * (unsigned)(a.in_ld * 2) + (unsigned)(kc * ROWB) +
                                            (unsigned)((slot ^ swz<BK>(row)) << 4)
                                      : GG_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, lds, 16, off, 0, 0, 0);
      } else {
        const int r = row - BM, n = n0 + r;
        const unsigned off = n < a.cout_total
                                 ? ((unsigned)(tap * a.cout_total + n) * (unsigned)a.cin) * 2u + (unsigned)(kc * ROWB) +
                                       (unsigned)((slot ^ swz<BK>(r)) << 4)
                                 : GG_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, lds, 16, off, 0, 0, 0);
      }
    }
  };

  auto issue = [&](int s) {
    const int tap = s / kchunks, kc = s - tap * kchunks;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ld_ok[j]) {
        if (ld_act[j]) {
          const int src = nbr_lds[tap * BM + ld_row[j]];
          if (src >= 0)
            v = *reinterpret_cast<const uint4*>(a.in + (size_t)src * a.in_ld + kc * BK + ld_chunk[j] * 8);
        } else {
          const int n = n0 + ld_row[j];
          if (n < a.cout_total)
            v = *reinterpret_cast<const uint4*>(a.wt + ((size_t)tap * a.cout_total + n) * a.cin + kc * BK +
                                                ld_chunk[j] * 8);
        }
      }
      regs[j] = v;
    }
  };
  auto commit = [&](int buf) {
    char* base = smem + buf * BUF_BYTES;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      if (!ld_ok[j]) continue;
      const int row = ld_row[j];
      char* dst = base + (ld_act[j] ? 0 : BM * ROWB) + row * ROWB + ((ld_chunk[j] ^ swz<BK>(row)) << 4);
      *reinterpret_cast<uint4*>(dst) = regs[j];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // NBUF == 3: the DMA of stage s+2 is issued while stage s computes and only stage s+1 is waited for
  // (counted vmcnt, raw barrier: a __syncthreads() would drain everything), so a stage never waits on
  // loads issued inside itself.
  constexpr int PIECES_PER_WAVE = ((BM + BN) / ROWS_PER_PIECE) / (NT / 64);
  static_assert(NBUF == 2 || ((BM + BN) / ROWS_PER_PIECE) % (NT / 64) == 0, "ring needs equal pieces per wave");
  if constexpr (DMA) {
    issue_dma(0, 0);
    if constexpr (NBUF == 3) {
      if (stages > 1) issue_dma(1, 1);
      if (stages > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES_PER_WAVE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {
    issue(0);
    commit(0);
  }
  __syncthreads();

  for (int s = 0; s < stages; ++s) {
    GG_STAMP(0);
    if constexpr (DMA) {
      if (s + NBUF - 1 < stages && !GG_DIAG(2)) issue_dma(s + NBUF - 1, (s + NBUF - 1) % NBUF);
    } else {
      if (s + 1 < stages) issue(s + 1);
    }
    GG_STAMP(1);
    const char* abase = smem + (s % NBUF) * BUF_BYTES;
    const char* wbase = abase + BM * ROWB;
    if (!GG_DIAG(1))
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[TM], fw[TN];
      const int chunk = 2 * ks + lh;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 32 + lr;
        fa[i] = *reinterpret_cast<const bf16x8*>(abase + row * ROWB + ((chunk ^ swz<BK>(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WTN + j * 32 + lr;
        fw[j] = *reinterpret_cast<const bf16x8*>(wbase + row * ROWB + ((chunk ^ swz<BK>(row)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
    }
    GG_STAMP(2);
    if constexpr (DMA && NBUF == 3) {
      // stage s+1 must have landed; stage s+2 (if issued) may stay in flight
      if (s + 2 < stages) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PIECES_PER_WAVE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      GG_STAMP(3);
      __syncthreads();
      GG_STAMP(4);
    } else {
      if (s + 1 < stages) commit((s + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue: D row = output channel (reg&3) + 8*(reg>>2) + 4*lh, D col = site lr -------------
  TOut* out = reinterpret_cast<TOut*>(a.out);
  constexpr bool LDS_EPILOGUE = sizeof(TOut) == 2 && (size_t)BM * (BN * 2 + 8) <= (size_t)NBUF * BUF_BYTES;
  if constexpr (LDS_EPILOGUE) {
    // bf16: finish the math per lane (scale/shift, residual, ReLU), park the tile in LDS, then write
    // whole output rows: 64 lanes x 8 B = 512 contiguous bytes per store instead of 32 rows x 16 B.
    constexpr int EROW = BN * 2 + 8;                 // +8 B: lanes of a wave walk rows, keep banks apart
    char* et = smem;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rl = wm * WTM + i * 32 + lr;
      const int m = m0 + rl;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cl = wn * WTN + j * 32 + 8 * g + 4 * lh;
          const int c = n0 + cl;
          f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
          if (c < a.cout_total) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + c);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + c);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * g + r];
            v = v * sc + sh;
            if (a.residual && m < M) {
              const bf16x4 rv = *reinterpret_cast<const bf16x4*>(a.residual + (size_t)m * a.res_ld + c);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
            }
            if (a.relu) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
          }
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (__bf16)v[r];
          *reinterpret_cast<bf16x4*>(et + rl * EROW + cl * 2) = o;
        }
      }
    }
    __syncthreads();
    constexpr int LPR = BN / 4;                      // lanes per output row (4 channels each)
    constexpr int RPI = 64 / LPR > 0 ? 64 / LPR : 1; // rows per wave-instruction
    static_assert(LPR <= 64, "row wider than a wavefront");
    const int sub = lane / LPR, cl = (lane % LPR) * 4;
    for (int r0 = wave * RPI; r0 < BM; r0 += (NT / 64) * RPI) {
      const int rl = r0 + sub, m = m0 + rl;
      if (rl < BM && m < M && n0 + cl < a.cout_total) {
        const size_t orow = a.out_rows ? (size_t)a.out_rows[m] : (size_t)m;
        const uint2 v = *reinterpret_cast<const uint2*>(et + rl * EROW + cl * 2);
        *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(out) + orow * a.out_ld + n0 + cl) = v;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * WTM + i * 32 + lr;
      if (m >= M) continue;
      const size_t orow = a.out_rows ? (size_t)a.out_rows[m] : (size_t)m;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = n0 + wn * WTN + j * 32 + 8 * g + 4 * lh;
          if (c >= a.cout_total) continue;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + c);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + c);
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * g + r];
          v = v * sc + sh;
          if (a.residual) {
            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(a.residual + (size_t)m * a.res_ld + c);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
          }
          if (a.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          OutOps<TOut>::store4(out + orow * a.out_ld + c, v);
        }
      }
    }
  }
}

template <int BM, int BN, int NWM, int NWN, int BK, typename TOut, int NBUF = 2>
static int launch_gg(const GGArgs& a, int groups, hipStream_t stream) {
  constexpr int NT = NWM * NWN * 64;
  constexpr size_t LDS2 = 2 * (size_t)(BM + BN) * BK * 2 + (size_t)GG_MAX_TAPS * BM * 4;
  constexpr size_t LDS3 = (size_t)NBUF * (size_t)(BM + BN) * BK * 2 + (size_t)GG_MAX_TAPS * BM * 4;
  static const bool no_dma = getenv("PN_GG_NO_DMA") != nullptr;
  static const bool no_ring = getenv("PN_GG_NO_RING") != nullptr;
  const bool dma = a.in_bytes != 0 && !no_dma;
  const bool ring = dma && NBUF == 3 && !no_ring;
  auto kern = ring ? gather_gemm_kernel<BM, BN, NWM, NWN, BK, TOut, true, NBUF>
                   : (dma ? gather_gemm_kernel<BM, BN, NWM, NWN, BK, TOut, true, 2>
                          : gather_gemm_kernel<BM, BN, NWM, NWN, BK, TOut, false, 2>);
  const size_t LDS = ring ? LDS3 : LDS2;
  static bool attr_set[3] = {false, false, false};
  const int which = ring ? 2 : (dma ? 1 : 0);
  if (!attr_set[which] && LDS > 32 * 1024) {
    PN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    attr_set[which] = true;
  }
  const dim3 grid(pn_div_up(a.rows, BM), pn_div_up(a.cout_total, BN), groups);
  hipLaunchKernelGGL(kern, grid, dim3(NT), LDS, stream, a);
  PN_CHECK_LAUNCH("gather_gemm_kernel");
  return PN_OK;
}

template <typename TOut>
static int dispatch_gg(const GGArgs& a, int groups, hipStream_t stream) {
  const int N = a.cout_total, rows = a.rows;
  static const char* force = getenv("PN_GG_FORCE");   // tuning aid: A / B / E for wide outputs
  if (a.cin % 64 != 0) {  // 32 input channels: one 32-wide stage per tap
    if (N <= 32) return launch_gg<128, 32, 4, 1, 32, TOut>(a, groups, stream);
    return launch_gg<128, 64, 2, 2, 32, TOut>(a, groups, stream);
  }
  if (N <= 32) return launch_gg<128, 32, 4, 1, 64, TOut>(a, groups, stream);
  if (N <= 64) return launch_gg<128, 64, 2, 2, 64, TOut>(a, groups, stream);
  if (force && N > 128) {
    if (force[0] == 'A') return launch_gg<128, 256, 2, 4, 64, TOut>(a, groups, stream);
    if (force[0] == 'B') return launch_gg<128, 128, 2, 2, 64, TOut>(a, groups, stream);
    if (force[0] == 'E') return launch_gg<64, 128, 2, 2, 64, TOut>(a, groups, stream);
    if (force[0] == 'F') return launch_gg<64, 256, 1, 4, 64, TOut>(a, groups, stream);
    if (force[0] == 'G') return launch_gg<256, 256, 4, 4, 64, TOut>(a, groups, stream);
    if (force[0] == 'R') return launch_gg<128, 256, 2, 4, 64, TOut, 3>(a, groups, stream);
  }
  if (N <= 128) {
    if (rows * (long long)groups >= 128 * 192) return launch_gg<128, 128, 2, 2, 64, TOut>(a, groups, stream);
    return launch_gg<64, 128, 2, 2, 64, TOut>(a, groups, stream);
  }
  // wide outputs: one 128 x 256 tile per workgroup when that still fills the chip, else split
  if (rows >= 128 * 192) return launch_gg<128, 256, 2, 4, 64, TOut>(a, groups, stream);
  if (rows >= 64 * 96 * 2) return launch_gg<128, 128, 2, 2, 64, TOut>(a, groups, stream);
  return launch_gg<64, 128, 2, 2, 64, TOut>(a, groups, stream);
}

static unsigned gg_in_bytes(long long in_rows, int in_ld) {
  const long long b = in_rows * (long long)in_ld * 2;
  return (in_rows > 0 && b < (long long)GG_OOB) ? (unsigned)b : 0u;   // 0 -> register staging
}

thread_local int pn_gg_free_order = 0;

extern "C" int pn_gather_gemm(const void* in, int cin, int in_ld, long long in_rows, const int32_t* nbr, int nbr_stride, int ntaps,
                              const int32_t* count_dev, int rows, const void* weight, int cout,
                              const float* scale, const float* shift, const void* residual,
                              int res_ld, int relu, void* out, int out_ld, int out_fp32,
                              const int32_t* out_rows, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(in && weight && scale && shift && out, "pn_gather_gemm: null argument");
  PN_CHECK_ARG(rows >= 1 && ntaps >= 1 && (nbr || ntaps == 1), "pn_gather_gemm: bad rows / taps");
  if (in_ld <= 0) in_ld = cin;
  PN_CHECK_ARG(in_ld >= cin && in_ld % 8 == 0, "pn_gather_gemm: in_ld %d (cin %d)", in_ld, cin);
  PN_CHECK_ARG(cin % 32 == 0 && cout % 4 == 0, "pn_gather_gemm: cin %d must be a multiple of 32, cout %d of 4",
               cin, cout);
  PN_CHECK_ARG(!nbr || nbr_stride >= rows, "pn_gather_gemm: rulebook stride %d < rows %d", nbr_stride, rows);
  PN_CHECK_ARG(out_ld % 4 == 0 && (!residual || res_ld % 4 == 0), "pn_gather_gemm: row strides must be multiples of 4");
  PN_CHECK_ARG((((uintptr_t)in | (uintptr_t)weight | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0 &&
                   ((uintptr_t)out & 7) == 0 && ((uintptr_t)residual & 7) == 0,
               "pn_gather_gemm: misaligned pointer");
  GGArgs a;
  a.in = (const __bf16*)in; a.in_ld = in_ld; a.in_bytes = gg_in_bytes(in_rows, in_ld); a.nbr = nbr;
  a.nbr_stride = nbr_stride; a.count = count_dev; a.rows = rows;
  a.wt = (const __bf16*)weight; a.cin = cin; a.cout_total = cout; a.ntaps = ntaps;
  a.scale = scale; a.shift = shift; a.residual = (const __bf16*)residual; a.res_ld = res_ld; a.relu = relu;
  a.out = out; a.out_ld = out_ld; a.out_rows = out_rows;
  a.g_in = a.g_wt = a.g_ss = a.g_out = a.g_rows = a.g_nbr = a.g_count = 0;
  { const char* d = getenv("PN_GG_DIAG"); a.diag = d ? atoi(d) : 0; }
  a.free_order = pn_gg_free_order;
  PN_CHECK_ARG(ntaps <= GG_MAX_TAPS, "pn_gather_gemm: at most %d taps", GG_MAX_TAPS);
  {
    const int rc = pn_gg_try_direct(a, 1, out_fp32 != 0, stream);   // 32 -> 32: no staging at all
    if (rc != -1) return rc;
  }
  {
    const int rc = pn_gg_try_window(a, 1, out_fp32 != 0, stream);   // 3x3 rulebooks: window-staged kernel
    if (rc != -1) return rc;
  }
  return out_fp32 ? dispatch_gg<float>(a, 1, stream) : dispatch_gg<__bf16>(a, 1, stream);
}

// `groups` independent problems that share rows / rulebook and differ by a channel offset of the
// input (group g reads in + g*in_group_cols), their own [ntaps, cout, cin] weights and scale/shift,
// and write out + g*out_group_cols: the per-task convs of CenterHead in one launch.
static int gather_gemm_grouped_impl(const void* in, int cin, int in_ld, long long in_rows, long long in_group_stride,
                                    const int32_t* nbr, int nbr_stride, long long nbr_group_stride, int ntaps,
                                    const int32_t* count_dev, int rows, const void* weight, int cout, int groups, const float* scale,
                                    const float* shift, int relu, void* out, int out_ld,
                                    long long out_group_stride, int out_fp32, const int32_t* out_rows,
                                    long long out_rows_group_stride, pnStream stream_);

extern "C" int pn_gather_gemm_grouped(const void* in, int cin, int in_ld, long long in_rows, int in_group_cols,
                                      const int32_t* nbr, int nbr_stride, int ntaps, int rows,
                                      const void* weight, int cout, int groups, const float* scale,
                                      const float* shift, int relu, void* out, int out_ld,
                                      int out_group_cols, int out_fp32, const int32_t* out_rows,
                                      long long out_rows_group_stride, pnStream stream_) {
  return gather_gemm_grouped_impl(in, cin, in_ld, in_rows, in_group_cols, nbr, nbr_stride, 0, ntaps, nullptr, rows, weight,
                                  cout, groups, scale, shift, relu, out, out_ld, out_group_cols, out_fp32, out_rows,
                                  out_rows_group_stride, stream_);
}

// Grouped form whose groups also differ in their rulebook and row block: group g reads rows through
// nbr + g*nbr_group_stride from in + g*in_group_stride (elements; 0 = every group gathers from the same
// rows) and writes out + g*out_group_stride (elements); counts_dev[g] (optional, device) = rows in use.
extern "C" int pn_gather_gemm_blocks(const void* in, int cin, int in_ld, long long in_rows, long long in_group_stride,
                                     const int32_t* nbr, int nbr_stride, long long nbr_group_stride, int ntaps,
                                     const int32_t* counts_dev, int rows, const void* weight, int cout, int groups, const float* scale,
                                     const float* shift, int relu, void* out, int out_ld,
                                     long long out_group_stride, int out_fp32, pnStream stream_) {
  PN_CHECK_ARG(in_group_stride % 8 == 0 && out_group_stride % 4 == 0, "pn_gather_gemm_blocks: group strides not aligned");
  return gather_gemm_grouped_impl(in, cin, in_ld, in_rows, in_group_stride, nbr, nbr_stride, nbr_group_stride, ntaps,
                                  counts_dev, rows, weight, cout, groups, scale, shift, relu, out, out_ld, out_group_stride,
                                  out_fp32, nullptr, 0, stream_);
}

static int gather_gemm_grouped_impl(const void* in, int cin, int in_ld, long long in_rows, long long in_group_cols,
                                    const int32_t* nbr, int nbr_stride, long long nbr_group_stride, int ntaps,
                                    const int32_t* count_dev, int rows, const void* weight, int cout, int groups, const float* scale,
                                    const float* shift, int relu, void* out, int out_ld,
                                    long long out_group_cols, int out_fp32, const int32_t* out_rows,
                                    long long out_rows_group_stride, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(in && weight && scale && shift && out && groups >= 1 && rows >= 1,
               "pn_gather_gemm_grouped: bad argument");
  PN_CHECK_ARG(ntaps >= 1 && ntaps <= GG_MAX_TAPS && (nbr || ntaps == 1), "pn_gather_gemm_grouped: bad taps");
  PN_CHECK_ARG(cin % 32 == 0 && cout % 4 == 0 && in_ld % 8 == 0 && in_group_cols % 8 == 0 && out_ld % 4 == 0 &&
                   out_group_cols % 4 == 0,
               "pn_gather_gemm_grouped: channel counts / strides not aligned");
  GGArgs a;
  a.in = (const __bf16*)in; a.in_ld = in_ld; a.in_bytes = gg_in_bytes(in_rows, in_ld); a.nbr = nbr;
  a.nbr_stride = nbr_stride; a.count = count_dev;
  a.rows = rows; a.wt = (const __bf16*)weight; a.cin = cin; a.cout_total = cout; a.ntaps = ntaps;
  a.scale = scale; a.shift = shift; a.residual = nullptr; a.res_ld = 0; a.relu = relu;
  a.out = out; a.out_ld = out_ld; a.out_rows = out_rows;
  a.g_in = in_group_cols; a.g_wt = (long long)ntaps * cout * cin; a.g_ss = cout; a.g_out = out_group_cols;
  a.g_rows = out_rows_group_stride;
  a.g_nbr = nbr_group_stride;
  a.g_count = count_dev ? 1 : 0;
  a.diag = 0;
  a.free_order = 0;
  {
    const int rc = pn_gg_try_window(a, groups, out_fp32 != 0, stream);
    if (rc != -1) return rc;
  }
  return out_fp32 ? dispatch_gg<float>(a, groups, stream) : dispatch_gg<__bf16>(a, groups, stream);
}

// out[p, c] = bias[c] + sum_tap z[nbr[tap][p], tap*C + c]   (fp32): the spatial half of a 3x3 conv
// whose per-tap channel products were computed first as one 1x1 GEMM (z = x . [W_0 | ... | W_8]).
template <typename TZ> __device__ __forceinline__ f32x4 load_z4(const TZ* p);
template <> __device__ __forceinline__ f32x4 load_z4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 load_z4<__bf16>(const __bf16* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

template <typename TZ>
__global__ __launch_bounds__(256) void tap_sum_kernel(const TZ* __restrict__ z, int z_ld,
                                                      const int32_t* __restrict__ nbr, int nbr_stride,
                                                      int rows, int C, const float* __restrict__ bias,
                                                      float* __restrict__ out, int out_ld,
                                                      long long g_z, long long g_out, int g_bias,
                                                      long long g_nbr) {
  const int g = blockIdx.y;
  z += g * g_z; out += g * g_out; bias += g * g_bias; nbr += g * g_nbr;
  const int c4 = C / 4;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < (long long)rows * c4;
       t += (long long)gridDim.x * 256) {
    const int p = (int)(t / c4), c = (int)(t - (long long)p * c4) * 4;
    f32x4 acc = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int src = nbr[(size_t)tap * nbr_stride + p];
      if (src >= 0) acc += load_z4<TZ>(z + (size_t)src * z_ld + tap * C + c);
    }
    *reinterpret_cast<f32x4*>(out + (size_t)p * out_ld + c) = acc;
  }
}

static int tap_sum_impl(const void* z, int z_dtype, int z_ld, long long z_group_stride, const int32_t* nbr,
                        int nbr_stride, long long nbr_group_stride, int rows, int channels, int groups,
                        const float* bias, float* out, int out_ld, long long out_group_stride, pnStream stream_);

extern "C" int pn_tap_sum(const void* z, int z_dtype, int z_ld, long long z_group_stride, const int32_t* nbr,
                          int nbr_stride, int rows, int channels, int groups, const float* bias,
                          float* out, int out_ld, long long out_group_stride, pnStream stream_) {
  return tap_sum_impl(z, z_dtype, z_ld, z_group_stride, nbr, nbr_stride, 0, rows, channels, groups, bias, out, out_ld,
                      out_group_stride, stream_);
}

// pn_tap_sum with one rulebook per group (nbr + g*nbr_group_stride)
extern "C" int pn_tap_sum_blocks(const void* z, int z_dtype, int z_ld, long long z_group_stride, const int32_t* nbr,
                                 int nbr_stride, long long nbr_group_stride, int rows, int channels, int groups,
                                 const float* bias, float* out, int out_ld, long long out_group_stride,
                                 pnStream stream_) {
  return tap_sum_impl(z, z_dtype, z_ld, z_group_stride, nbr, nbr_stride, nbr_group_stride, rows, channels, groups,
                      bias, out, out_ld, out_group_stride, stream_);
}

static int tap_sum_impl(const void* z, int z_dtype, int z_ld, long long z_group_stride, const int32_t* nbr,
                        int nbr_stride, long long nbr_group_stride, int rows, int channels, int groups,
                        const float* bias, float* out, int out_ld, long long out_group_stride, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(z && nbr && bias && out && rows >= 1 && groups >= 1, "pn_tap_sum: null argument");
  PN_CHECK_ARG(channels % 4 == 0 && z_ld % 4 == 0 && out_ld % 4 == 0 && z_ld >= 9 * channels,
               "pn_tap_sum: channels / strides must be multiples of 4");
  int bx = pn_div_up((long long)rows * (channels / 4), 256);
  if (bx > 2048) bx = 2048;
  if (z_dtype == PN_DTYPE_BF16)
    hipLaunchKernelGGL(tap_sum_kernel<__bf16>, dim3(bx, groups), dim3(256), 0, stream, (const __bf16*)z, z_ld, nbr,
                       nbr_stride, rows, channels, bias, out, out_ld, z_group_stride, out_group_stride, channels,
                       nbr_group_stride);
  else
    hipLaunchKernelGGL(tap_sum_kernel<float>, dim3(bx, groups), dim3(256), 0, stream, (const float*)z, z_ld, nbr,
                       nbr_stride, rows, channels, bias, out, out_ld, z_group_stride, out_group_stride, channels,
                       nbr_group_stride);
  PN_CHECK_LAUNCH("tap_sum_kernel");
  return PN_OK;
}
