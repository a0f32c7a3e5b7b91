// Window-staged gather -> MFMA implicit GEMM for 3x3 rulebooks (bf16 in, fp32 accumulate), gfx950.
//
// Same contract as gather_gemm.hip (out[m] = epilogue(sum_tap W[tap] . in[nbr[tap][m]])), different
// staging.  Pillar rows are sorted by flat BEV index and a tile is BM *consecutive* output sites, so
// the input rows that the three taps of one kernel row ky touch -- cells (y+ky-1, x-1..x+1) of every
// site of the tile -- are one nearly contiguous run [lo, hi] of the input row list (exactly BM+2 rows
// on a dense grid).  Per (ky, channel chunk) the workgroup therefore copies that run into LDS ONCE, as
// a linear block (no per-row indices, full 128-B lines), and the three taps kx = 0..2 read their MFMA
// operand rows out of it through a small per-site row table.  Against staging BM gathered rows per
// tap this cuts the activation traffic from L2 by ~2.7x (9*BM -> 3*(BM+2) rows per chunk); the
// per-CU L2 request rate is what bounds the per-tap kernel.
//
// Per tile the kernel checks that all three runs fit the window (WMAX rows); a tile where one does
// not (a rulebook without this locality, or neighbouring image rows much denser than the tile's) is
// processed by the same pipeline in "gather mode": the window is filled with the BM gathered rows of
// one tap and the row table is the identity.  Results are identical in both modes.
//
// Wave specialisation.  In-kernel s_memtime stamps of the per-tap kernel (scripts/gg_stamps.py) show a
// wave spending ~1000 of a stage's ~2900 cycles ISSUING its six LDS-DMA pieces (165 cycles each),
// ~850 on its 16 MFMAs and ~800 waiting (vmcnt + barrier) -- and every wave of a SIMD is in the same
// phase at the same time.  Here NWC compute waves (one per SIMD, 64 x 128 or 64 x 64 tiles) only read
// LDS and issue MFMAs, and NWL loader waves only compute addresses and issue DMA: weights run two
// stages ahead through a 3-deep ring (`s_waitcnt vmcnt(N)` leaves the newest stage in flight), the
// window one unit ahead.  One `s_barrier` per stage hands buffers between the two roles.
#include "gg_common.h"
#include <stdlib.h>

// The loader waves share their SIMD's issue port with the compute waves' MFMAs (a wave64 VALU op holds
// it for 4 cycles, an MFMA for 8 of its 32): ~30 address instructions per DMA piece -- what the first
// version of this kernel and the per-tap kernel spend -- cost more issue time than the MFMAs themselves.
// So everything per-lane is computed ONCE (LoaderLanes) and a piece is `s_mov m0` + one buffer_load whose
// stage-dependent part of the address travels in the scalar offset operand.
template <int NPW, int NPA>
struct LoaderLanes {
  unsigned w[NPW];   // weight piece j: ((n0 + row) * cin) * 2 + swizzled chunk, or GG_OOB past cout
  unsigned a[NPA];   // window piece j: row * in_row_bytes + swizzled chunk (row = piece row inside the window)
  int arow[NPA];     // ... and that row, for the `row < cnt` test of a run's last piece
};

// Window fill -> `base`.  Window mode: rows lo .. lo+cnt-1 of the input as a linear block (pieces past
// the run are skipped: stale LDS there is never addressed).  Gather mode (slow path): the BM rows
// nbr_row[0..BM).  Lane l of a piece fills slot l%CH of row l/CH and fetches chunk (slot ^ swizzle(row))
// of the source row, so the LDS image stays linear and the MFMA operand reads are bank-conflict free.
// ALL: every wave issues exactly NPA pieces (a piece past the run re-zeroes the zero rows), so that a counted
// `s_waitcnt vmcnt(NPA)` can leave a window fill in flight across a barrier.
template <int BK, int WMAX, int BM, int NW, int NPW, int NPA, bool ALL = false>
__device__ __forceinline__ void win_fill(char* base, __amdgpu_buffer_rsrc_t rs_in, int wave, int lane, bool win,
                                         int lo, int cnt, int kc, const int* nbr_row, unsigned in_row_bytes,
                                         const LoaderLanes<NPW, NPA>& L) {
  constexpr int CH = BK / 8, ROWB = BK * 2, RPP = 1024 / ROWB;
  if (win) {
    const unsigned soff = (unsigned)lo * in_row_bytes + (unsigned)(kc * ROWB);   // scalar
    if constexpr (ALL) {
      // every piece is issued, without a branch (a scalar branch per piece costs the issuing wave ~20 cycles, and
      // in a fill stage the staging waves are the ones the barrier waits for): a piece past the run reads out of
      // range -- zeros -- onto the (always zero) rows behind the window
#pragma unroll
      for (int j = 0; j < NPA; ++j) {
        const int piece = wave + j * NW;
        const bool in_run = piece * RPP < cnt;
        const unsigned off = L.arow[j] < cnt ? L.a[j] : GG_OOB;   // (arow >= piece * RPP: out of range whenever the piece is)
        auto lds = (__attribute__((address_space(3))) void*)(base + (in_run ? piece * 1024 : WMAX * ROWB));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, lds, 16, off, soff, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NPA; ++j) {
        const int piece = wave + j * NW;
        if (piece * RPP >= cnt) break;
        const unsigned off = L.arow[j] < cnt ? L.a[j] : GG_OOB;
        auto lds = (__attribute__((address_space(3))) void*)(base + piece * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, lds, 16, off, soff, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      const int piece = wave + j * NW;
      if (piece * RPP >= BM) break;
      const int row = piece * RPP + lane / CH, slot = lane % CH;
      const int src = nbr_row[row];
      const unsigned off = src >= 0 ? (unsigned)src * in_row_bytes + (unsigned)(kc * ROWB) +
                                          (unsigned)((slot ^ swz<BK>(row)) << 4)
                                    : GG_OOB;
      auto lds = (__attribute__((address_space(3))) void*)(base + piece * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, lds, 16, off, 0, 0, 0);
    }
  }
}

// weight tile (tap, channel chunk kc), BN rows -> `base`: NPW pieces per loader wave, no per-piece VALU
template <int BK, int BN, int NW, int NPW, int NPA>
__device__ __forceinline__ void win_wload(char* base, __amdgpu_buffer_rsrc_t rs_wt, int wave, int tap, int kc,
                                          int cout_total, int cin, const LoaderLanes<NPW, NPA>& L) {
  constexpr int ROWB = BK * 2;
  const unsigned soff = (unsigned)(tap * cout_total) * (unsigned)cin * 2u + (unsigned)(kc * ROWB);   // scalar
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int piece = wave + j * NW;
    auto lds = (__attribute__((address_space(3))) void*)(base + piece * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, lds, 16, L.w[j], soff, 0, 0);
  }
}

// Probe builds only (-DGGW_STAMPS, scripts/build_probe_lib.sh; the shipped kernels carry no stamp code).
// diag == 4 + 16 * b: s_memtime stamps of workgroup (b, 0) (read with pn_gg_win_debug_stamps):
// [0..3] kernel start / prologue done / main loop done / end (compute wave 0); then per stage 4 slots:
// compute wave 0 {stage start, MFMAs done, barrier passed}, loader wave 0 {ready for the barrier};
// then 8 prologue / epilogue marks
static constexpr int GGW_STAMP_STAGES = 40;
__device__ unsigned long long ggw_stamps[4 + GGW_STAMP_STAGES * 4 + 8];
// (the same mode records every workgroup's start and end tick, to see how a launch's workgroups overlap in time)
static constexpr int GGW_SPAN_BLOCKS = 2048;
__device__ unsigned long long ggw_spans[GGW_SPAN_BLOCKS * 2];
// (likewise the probes that drop the MFMAs, diag 1, or the staging, diag 2: a scalar branch per stage costs the
// stage loop ~20 cycles)
#ifdef GGW_STAMPS
#define GGW_DIAG(x) (a.diag == (x))
#else
#define GGW_DIAG(x) false
#endif
#ifdef GGW_STAMPS
#define GGW_STAMP(cond, idx)                                                    \
  do {                                                                          \
    if ((a.diag & 15) == 4 && (cond)) {                                         \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();               \
      if ((int)blockIdx.x == (a.diag >> 4) && blockIdx.y == 0 && lane == 0) ggw_stamps[(idx)] = t_; \
      const int lb_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
      if (((idx) == 0 || (idx) == 3) && lb_ < GGW_SPAN_BLOCKS && lane == 0) ggw_spans[2 * lb_ + ((idx) == 3)] = t_; \
    }                                                                           \
  } while (0)
#else
#define GGW_STAMP(cond, idx) do {} while (0)
#endif
// (and, for the stamped workgroup, every wave's arrival at / departure from the stage barrier of stages
// GGW_WAVE_S0 .. +3 in the per-tap loop: [wave][stage][arrive, leave, stage start])
static constexpr int GGW_WAVE_S0 __attribute__((unused)) = 8;
__device__ unsigned long long ggw_wave[16 * 4 * 3 + 16 * 4];
#ifdef GGW_STAMPS
#define GGW_WSTAMP(s_, which)                                                                  \
  do {                                                                                         \
    if ((a.diag & 15) == 4 && (int)blockIdx.x == (a.diag >> 4) && blockIdx.y == 0 && (s_) >= GGW_WAVE_S0 && \
        (s_) < GGW_WAVE_S0 + 4 && wave < 16) {                                                 \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                              \
      if (lane == 0) ggw_wave[(wave * 4 + ((s_)-GGW_WAVE_S0)) * 3 + (which)] = t_;             \
    }                                                                                          \
  } while (0)
#else
#define GGW_WSTAMP(s_, which) do {} while (0)
#endif
// -DGGW_KSTAMPS (probe build only): the tick at which each k-step of stage GGW_WAVE_S0 + 2 starts (fragments
// landed), per wave of the stamped workgroup, behind the barrier stamps
#ifdef GGW_KSTAMPS
#define GGW_KSTAMP(ks_)                                                                        \
  do {                                                                                         \
    if ((a.diag & 15) == 4 && (int)blockIdx.x == (a.diag >> 4) && blockIdx.y == 0 && kstamp_stage == GGW_WAVE_S0 + 2 && \
        wave < 16 && (ks_) < 4) {                                                              \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                              \
      if (lane == 0) ggw_wave[16 * 4 * 3 + wave * 4 + (ks_)] = t_;                             \
    }                                                                                          \
  } while (0)
#else
#define GGW_KSTAMP(ks_) do {} while (0)
#endif
extern "C" int pn_gg_win_debug_waves(unsigned long long* host_out, int n) {
  if (n > 16 * 4 * 3 + 16 * 4) n = 16 * 4 * 3 + 16 * 4;
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ggw_wave), (size_t)n * 8);
}
extern "C" int pn_gg_win_debug_spans(unsigned long long* host_out, int n) {
  if (n > GGW_SPAN_BLOCKS * 2) n = GGW_SPAN_BLOCKS * 2;
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ggw_spans), (size_t)n * 8);
}
extern "C" int pn_gg_win_debug_stamps(unsigned long long* host_out, int n) {
  const int total = 4 + GGW_STAMP_STAGES * 4 + 8;
  if (n > total) n = total;
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ggw_stamps), (size_t)n * 8);
}

// LDS pool = the staging buffers (2 windows + 3 weight tiles), or the parked fp32 output tile if larger
template <int BM, int BN, int BK, int WMAX, bool UNIT, int RING, int EP = 1>
constexpr int win_pool_bytes() {
  constexpr int RPP = 1024 / (BK * 2);
  constexpr int staging = 2 * (WMAX + RPP) * BK * 2 + (UNIT ? 6 : RING) * BN * BK * 2;
  constexpr int parked = BM / EP * (BN * 4 + 16);      // EP: the tile leaves through the pool in EP passes of BM / EP rows
  return staging > parked ? staging : parked;
}

// UNIT (narrow layers, where a tap's weight tile is small): in window mode a stage is a whole kernel row --
// its three taps' weights land together, double-buffered -- so a tile has 3*kchunks barriers instead of
// 9*kchunks; a 16-MFMA stage spent more time on its barrier and first LDS reads than on its MFMAs.
// RING: weight tiles in flight + in use on the per-tap path: 3 (two stages ahead, counted vmcnt) or 2 (one
// ahead; the smaller pool lets two workgroups share a CU and cover each other's barriers instead).
// NTAPS: 9 (3x3 rulebook: three kernel rows of three taps) or 1 (a plain GEMM over the rows, rulebook optional --
// the "window" of a tile is then its own BM consecutive rows).
// EP: epilogue passes (the fp32 tile leaves through the pool BM / EP rows at a time).  FUSED (NWL = 0): no loader
// waves -- every wave issues its share of a stage's DMA pieces at the top of the stage and then computes; with
// 256 x 256 tiles a workgroup stages 43 pieces per 64 MFMAs per SIMD instead of 38 per 32, which is what the
// vector issue port of a SIMD is short of (a piece costs it ~150 cycles, an MFMA 8 of its 32).
template <int BM, int BN, int NWM, int NWN, int NWL, int BK, int WMAX, typename TOut, bool UNIT, int RING, int NTAPS, int EP,
          bool FUSED>
__global__ __launch_bounds__((NWM * NWN + NWL) * 64) void gg_win_kernel(GGArgs a, int force_gather) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass has no __amdgpu_buffer_rsrc_t to check the helpers against
  constexpr int NWC = NWM * NWN;              // compute waves; waves NWC .. NWC+NWL-1 are loaders
  constexpr int NT = (NWC + NWL) * 64, NW = NT / 64;
  constexpr int CH = BK / 8;                  // 16-B chunks per LDS row
  constexpr int ROWB = BK * 2;                // bytes per LDS row
  constexpr int RPP = 1024 / ROWB;            // rows one wave-wide 16-B DMA instruction lands
  constexpr int WTM = BM / NWM, WTN = BN / NWN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int ABUF = (WMAX + RPP) * ROWB;   // window + one piece of zero rows (row WMAX = "no neighbour")
  constexpr int WBUF = BN * ROWB;
  static_assert(NTAPS == 9 || NTAPS == 1, "3x3 rulebooks or 1-tap GEMMs");
  constexpr int TPG = NTAPS == 9 ? 3 : 1;     // taps per kernel row (= per window)
  constexpr int NG = NTAPS / TPG;             // kernel rows
  static_assert(!UNIT || NTAPS == 9, "unit stages are a 3x3 notion");
  static_assert(RING == 2 || RING == 3, "weight ring of 2 or 3");
  static_assert(!UNIT || RING == 3, "unit kernels keep the 3-deep ring for their gather-mode tiles");
  constexpr int POOL = win_pool_bytes<BM, BN, BK, WMAX, UNIT, RING, EP>();   // staging buffers; the epilogue parks the fp32 tile here
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be made of 32x32 MFMA tiles");
  static_assert(NWM * NWN + NWL >= NG, "one wave per kernel row reduces the rulebook tile");
  static_assert(WMAX % RPP == 0 && BM % RPP == 0 && BN % RPP == 0 && BM <= WMAX, "window geometry");
  // waves that issue DMA.  Fused: one of the two waves of every SIMD (waves 0-3) stages for both -- its partner's MFMAs
  // run under the issue burst, its own MFMAs under the partner's wait; with all eight issuing at the top of a stage
  // the bursts of a SIMD's two waves add up in front of the MFMAs of both (3 560 against 2 300 cycles per stage)
  constexpr int NLD = FUSED ? 4 : NWL;
  static_assert(FUSED == (NWL == 0) && (!FUSED || (!UNIT && RING == 2 && NTAPS == 9)), "fused waves: per-tap stages, 2-deep ring");
  static_assert((BN / RPP) % NLD == 0, "every loader wave must issue the same number of weight pieces");
  constexpr int NWP = BN / RPP / NLD;         // weight pieces per loader wave and stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wsm = smem + 2 * ABUF;
  int* const nbr_lds = reinterpret_cast<int*>(smem + POOL);   // [9][BM]
  int* const meta = nbr_lds + NTAPS * BM;                          // lo/hi per kernel row (8 ints)
  float* const sc_lds = reinterpret_cast<float*>(meta + 8);    // [BN] scale, [BN] shift

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
  int bx = blockIdx.x;
  {  // XCD-contiguous bands of row tiles (see gather_gemm.hip)
    const int nwg = (M + BM - 1) / BM;
    if (bx >= nwg) return;
    const int x = bx & 7, q = nwg >> 3, r = nwg & 7;
    bx = x * q + (x < r ? x : r) + (bx >> 3);
  }
  const int m0 = bx * BM;
  const int n0 = blockIdx.y * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int kchunks = a.cin / BK;
  const int nstages = NTAPS * kchunks;             // stage s = (kernel row g, channel chunk kc, kx t), t fastest
  const bool loader = wave >= NWC;             // wave-uniform role (never true with fused waves)
  const int lw = FUSED ? wave : wave - NWC;    // loader index (pieces are dealt round-robin over NLD)
  GGW_STAMP(wave == 0, 0);

  // the rulebook tile and scale / shift first: their round trip (the longest latency of the prologue) runs
  // under the descriptor / lane-offset setup and the first weight DMA below
  int nb[(NTAPS * BM + NT - 1) / NT];
#pragma unroll
  for (int j = 0; j < (NTAPS * BM + NT - 1) / NT; ++j) {   // all loads in flight before the first use
    const int q = tid + j * NT, tap = q / BM, r = q - tap * BM, m = m0 + r;
    nb[j] = (q < NTAPS * BM && m < M) ? (a.nbr ? a.nbr[(size_t)tap * a.nbr_stride + m] : m) : -1;
  }
  float scv = 0.f, shv = 0.f;
  if (tid < BN && n0 + tid < a.cout_total) {
    scv = a.scale[n0 + tid];
    shv = a.shift[n0 + tid];
  }

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.wt, 0, (int)((size_t)NTAPS * a.cout_total * a.cin * 2), 0x00020000);
  const unsigned in_row_bytes = (unsigned)a.in_ld * 2u;
  constexpr int NPA = (WMAX / RPP + NLD - 1) / NLD;   // window pieces per loader wave
  LoaderLanes<NWP, NPA> LL;
  {
    const int lrow = lane / CH, slot = lane % CH;
#pragma unroll
    for (int j = 0; j < NWP; ++j) {
      const int r = ((loader || FUSED ? lw : 0) + j * NLD) * RPP + lrow, n = n0 + r;
      LL.w[j] = n < a.cout_total ? ((unsigned)n * (unsigned)a.cin) * 2u + (unsigned)((slot ^ swz<BK>(r)) << 4) : GG_OOB;
    }
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      const int r = ((loader || FUSED ? lw : 0) + j * NLD) * RPP + lrow;
      LL.arow[j] = r;
      LL.a[j] = (unsigned)r * in_row_bytes + (unsigned)((slot ^ swz<BK>(r)) << 4);
    }
  }
#define WIN_WLOAD_SLOT(s_, slot_)                                                                         \
  do {                                                                                                    \
    const int u_ = (s_) / TPG, g_ = u_ / kchunks;                                                         \
    win_wload<BK, BN, NLD, NWP, NPA>(wsm + (slot_) * WBUF, rs_wt, lw, TPG * g_ + ((s_)-TPG * u_),          \
                                     u_ - g_ * kchunks, a.cout_total, a.cin, LL);                         \
  } while (0)
#define WIN_WLOAD(s_) WIN_WLOAD_SLOT(s_, (s_) % RING)

  // the weights do not depend on the rulebook: stages 0 and 1 are under way before anything else
  if (loader || (FUSED && wave < NLD)) {
    WIN_WLOAD(0);
    if (RING == 3 && nstages > 1) WIN_WLOAD(1);
  }

  // ---- rulebook tile -> LDS, run [lo, hi] of every kernel row, zero rows, scale/shift ------------
  if (tid < 2 * RPP * CH) {   // RPP zero rows in each of the two window buffers
    const int b = tid / (RPP * CH), q = tid - b * (RPP * CH);
    *reinterpret_cast<uint4*>(smem + b * ABUF + WMAX * ROWB + q * 16) = make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < (NTAPS * BM + NT - 1) / NT; ++j) {
    const int q = tid + j * NT;
    if (q < NTAPS * BM) nbr_lds[q] = nb[j];
  }
  if (tid < BN) {
    sc_lds[tid] = scv;
    sc_lds[BN + tid] = shv;
  }
  if (tid < 6) meta[tid] = (tid & 1) ? -1 : 0x7fffffff;   // empty runs for the kernel rows nobody reduces
  static_assert(BN <= NT, "one thread per channel stages scale / shift");
  GGW_STAMP(wave == 0, 4 + GGW_STAMP_STAGES * 4 + 0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (no vmcnt drain: the weight DMA stays in flight)
  if (wave < NG) {   // wave g reduces kernel row g: entries [TPG*g*BM, TPG*(g+1)*BM) of the tile
    int lo = 0x7fffffff, hi = -1;
#pragma unroll
    for (int j = 0; j < (TPG * BM + 63) / 64; ++j) {
      const int q = lane + j * 64;
      if (q < TPG * BM) {
        const int v = nbr_lds[wave * TPG * BM + q];
        if (v >= 0) {
          lo = min(lo, v);
          hi = max(hi, v);
        }
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      lo = min(lo, __shfl_xor(lo, d));
      hi = max(hi, __shfl_xor(hi, d));
    }
    if (lane == 0) {
      meta[2 * wave] = lo;
      meta[2 * wave + 1] = hi;
    }
  }
  GGW_STAMP(wave == 0, 4 + GGW_STAMP_STAGES * 4 + 1);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  GGW_STAMP(wave == 0, 4 + GGW_STAMP_STAGES * 4 + 2);
  // scalars, not an array: a dynamically indexed array would live in scratch
  const int h0 = __builtin_amdgcn_readfirstlane(meta[1]), h1 = __builtin_amdgcn_readfirstlane(meta[3]),
            h2 = __builtin_amdgcn_readfirstlane(meta[5]);
  const int lo0 = h0 >= 0 ? __builtin_amdgcn_readfirstlane(meta[0]) : 0, cnt0 = h0 >= 0 ? h0 - lo0 + 1 : 0;
  const int lo1 = h1 >= 0 ? __builtin_amdgcn_readfirstlane(meta[2]) : 0, cnt1 = h1 >= 0 ? h1 - lo1 + 1 : 0;
  const int lo2 = h2 >= 0 ? __builtin_amdgcn_readfirstlane(meta[4]) : 0, cnt2 = h2 >= 0 ? h2 - lo2 + 1 : 0;
  const bool win = !force_gather && cnt0 <= WMAX && cnt1 <= WMAX && cnt2 <= WMAX;
  // window mode: one fill per (g, kc) serves its three stages; gather mode: one fill (the BM rows of
  // the stage's tap) per stage.  Fill f goes to window buffer f & 1.
#define UNIT_SEL(gt, v0, v1, v2) ((gt) == 0 ? (v0) : ((gt) == 1 ? (v1) : (v2)))
#define WIN_FILL(f)                                                                                        \
  do {                                                                                                     \
    const int u_ = win ? (f) : (f) / TPG, g_ = u_ / kchunks, kc_ = u_ - g_ * kchunks;                      \
    const int tap_ = win ? 0 : TPG * g_ + ((f)-TPG * u_);                                                  \
    win_fill<BK, WMAX, BM, NLD, NWP, NPA, FUSED>(smem + ((f)&1) * ABUF, rs_in, lw, lane, win,              \
                                          UNIT_SEL(g_, lo0, lo1, lo2), win ? UNIT_SEL(g_, cnt0, cnt1, cnt2) : BM, \
                                          kc_, nbr_lds + tap_ * BM, in_row_bytes, LL);                     \
  } while (0)
  const int spf = win ? TPG : 1;               // stages per fill
  const int nfills = nstages / spf;
  // A stage's coordinates (t = tap inside the kernel row, kc = channel chunk, g = kernel row; t fastest) are
  // advanced, not derived from the stage index: the divisions by 3 and by the chunk count cost a wave ~25 scalar
  // instructions per stage, in front of the barrier every other wave waits at.
#define STAGE_NEXT(t1_, kc1_, g1_, t0_, kc0_, g0_)   \
  int t1_ = (t0_) + 1, kc1_ = (kc0_), g1_ = (g0_);    \
  if (t1_ == TPG) {                                   \
    t1_ = 0;                                          \
    if (++kc1_ == kchunks) {                          \
      kc1_ = 0;                                       \
      ++g1_;                                          \
    }                                                 \
  }
#define STAGE_ADV(t_, kc_, g_)      \
  do {                              \
    if (++(t_) == TPG) {            \
      (t_) = 0;                     \
      if (++(kc_) == kchunks) {     \
        (kc_) = 0;                  \
        ++(g_);                     \
      }                             \
    }                               \
  } while (0)
#define WIN_WLOAD_AT(slot_, g_, kc_, t_) \
  win_wload<BK, BN, NLD, NWP, NPA>(wsm + (slot_) * WBUF, rs_wt, lw, TPG * (g_) + (t_), (kc_), a.cout_total, a.cin, LL)
#define WIN_FILL_AT(par_, g_, kc_, t_)                                                                       \
  win_fill<BK, WMAX, BM, NLD, NWP, NPA, FUSED>(smem + (par_) * ABUF, rs_in, lw, lane, win,                   \
                                               UNIT_SEL(g_, lo0, lo1, lo2), win ? UNIT_SEL(g_, cnt0, cnt1, cnt2) : BM, \
                                               (kc_), nbr_lds + (win ? 0 : TPG * (g_) + (t_)) * BM, in_row_bytes, LL)

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool unit_mode = UNIT && win;
#ifdef GGW_KSTAMPS
  int kstamp_stage = -1;
#endif
  if constexpr (!FUSED) {
  if (loader) {
    WIN_FILL(0);
    if (unit_mode) {
      // ---- loader waves, unit stages: everything of unit u+1 (window + three weight tiles) goes out during
      // unit u and is waited for at its end; a unit is long enough (48+ MFMAs per wave) to cover the round trip
      WIN_WLOAD_SLOT(2, 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      int ukc = 0, ug = 0;   // this unit's channel chunk and kernel row
      for (int u = 0; u < nfills; ++u) {
        int kc1 = ukc + 1, g1 = ug;
        if (kc1 == kchunks) {
          kc1 = 0;
          ++g1;
        }
        if (u + 1 < nfills && !GGW_DIAG(2)) {
          WIN_FILL_AT((u + 1) & 1, g1, kc1, 0);
#pragma unroll
          for (int t = 0; t < 3; ++t) WIN_WLOAD_AT(((u + 1) & 1) * 3 + t, g1, kc1, t);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ukc = kc1, ug = g1;
      }
    } else {
      // ---- loader waves.  Per stage: the next fill (at the first stage of the current one) and the weights
      // two stages ahead; vmcnt(NWP) = "all but the weight pieces issued last have landed".
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // prologue only: stage 1's weights are waited too
      __builtin_amdgcn_s_barrier();
      int s = 0;
      for (int f = 0; f < nfills; ++f) {
        for (int t = 0; t < spf; ++t, ++s) {
          if (!GGW_DIAG(2)) {   // (stage indices divided out here: measured no slower than advancing them, the loaders are not on the critical path)
            if (t == 0 && f + 1 < nfills) WIN_FILL(f + 1);
            if (s + RING - 1 < nstages) WIN_WLOAD(s + RING - 1);
          }
          if (RING == 3 && s + 2 < nstages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWP) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          GGW_STAMP(lw == 0 && s < GGW_STAMP_STAGES, 4 + s * 4 + 3);   // loader 0 ready for the barrier
          GGW_WSTAMP(s, 0);
          __builtin_amdgcn_s_barrier();
          GGW_WSTAMP(s, 1);
        }
      }
    }
  } else {
    // ---- compute waves: LDS reads + MFMAs only
    const int wm = wave / NWN, wn = wave % NWN;
    int aoff[3][TM], akey[3][TM];   // window row (byte offset, swizzle key) of this lane's sites, per tap slot
#define ROW_TABLE(s_, slot_)                                                 \
  do {                                                                       \
    const int u_ = (s_) / TPG, g_ = u_ / kchunks, tap_ = TPG * g_ + ((s_)-TPG * u_); \
    const int lo_ = UNIT_SEL(g_, lo0, lo1, lo2);                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                         \
      const int site = wm * WTM + i * 32 + lr;                               \
      const int src = nbr_lds[tap_ * BM + site];                             \
      const int R = src >= 0 ? (win ? src - lo_ : site) : WMAX;              \
      aoff[slot_][i] = R * ROWB;                                             \
      akey[slot_][i] = swz<BK>(R);                                           \
    }                                                                        \
  } while (0)
    // the same in two halves: the rulebook entries of the next stage's tap are requested behind the current stage's
    // last fragment reads (they land under its last k-step), the table is finished after the last MFMA
#define ROW_LOAD(tap_, dst_)                                                 \
  do {                                                                       \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) dst_[i] = nbr_lds[(tap_) * BM + wm * WTM + i * 32 + lr]; \
  } while (0)
#define ROW_FINISH(src_, lo_, slot_)                                         \
  do {                                                                       \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                         \
      const int site = wm * WTM + i * 32 + lr;                               \
      const int R = src_[i] >= 0 ? (win ? src_[i] - (lo_) : site) : WMAX;    \
      aoff[slot_][i] = R * ROWB;                                             \
      akey[slot_][i] = swz<BK>(R);                                           \
    }                                                                        \
  } while (0)
#define DS_READ_FRAG(dst, base, key, ks) \
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"((base) + (unsigned)(((2 * (ks)) ^ (key)) << 4)))
    // Hand-ordered k-steps of one (tap, channel chunk) (inline asm; hipcc's own order exposed the LDS latency
    // once per k-step, ~500 of a stage's 1600 cycles, and sched_group_barrier made it chain MFMAs on one
    // accumulator).  Fragments are double-buffered: the reads of step ks+1 go out two per MFMA behind the
    // first MFMAs of step ks and are waited for (lgkmcnt counts only what this block issued) at the top of
    // step ks+1.  A wave tile of one or two MFMAs per k-step has no such cover: there all reads of step ks+1 go
    // out before the first MFMA of step ks (128 x 64 tiles on a 90 x 90 map: 19.1 -> 16.1 us, the 16-output
    // heat-map conv 23.5 -> 20.5; on four or more MFMAs per k-step the same order costs 5-8 registers and is
    // 0-5 % slower).
#define STAGE_MFMA(abase_, wbase_, slot_, pre_)                                                            \
  do {                                                                                                     \
    const unsigned a_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(abase_);      \
    const unsigned w_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(wbase_);      \
    unsigned ra[TM], rw[TN];                                                                               \
    int ka[TM], kw[TN];                                                                                    \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                       \
      ra[i] = a_lds + (unsigned)aoff[slot_][i];                                                            \
      ka[i] = akey[slot_][i] ^ lh;                                                                         \
    }                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                       \
      const int row_ = wn * WTN + j * 32 + lr;                                                             \
      rw[j] = w_lds + (unsigned)(row_ * ROWB);                                                             \
      kw[j] = swz<BK>(row_) ^ lh;                                                                          \
    }                                                                                                      \
    bf16x8 fa[2][TM], fw[2][TN];                                                                           \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) DS_READ_FRAG(fa[0][i], ra[i], ka[i], 0);                \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) DS_READ_FRAG(fw[0][j], rw[j], kw[j], 0);                \
    _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                                               \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
      GGW_KSTAMP(ks);                                                                                      \
      int nxt = 0;                                                                                         \
      if (TM * TN <= 2 && ks + 1 < BK / 16) {                                                           \
        _Pragma("unroll") for (; nxt < TM + TN; ++nxt) {                                                   \
          if (nxt < TM) DS_READ_FRAG(fa[(ks + 1) & 1][nxt], ra[nxt], ka[nxt], ks + 1);                     \
          else DS_READ_FRAG(fw[(ks + 1) & 1][nxt - TM], rw[nxt - TM], kw[nxt - TM], ks + 1);               \
        }                                                                                                  \
      }                                                                                                    \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) {      \
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"                                             \
                     : "+v"(acc[i][j])                                                                     \
                     : "v"(fw[ks & 1][j]), "v"(fa[ks & 1][i]));                                            \
        if (ks + 1 < BK / 16) {                                                                            \
          _Pragma("unroll") for (int r = 0; r < 2; ++r, ++nxt) {                                           \
            if (nxt < TM) DS_READ_FRAG(fa[(ks + 1) & 1][nxt], ra[nxt], ka[nxt], ks + 1);                   \
            else if (nxt < TM + TN)                                                                        \
              DS_READ_FRAG(fw[(ks + 1) & 1][nxt - TM], rw[nxt - TM], kw[nxt - TM], ks + 1);                \
          }                                                                                                \
        }                                                                                                  \
      }                                                                                                    \
      if (ks == (BK / 16 >= 2 ? BK / 16 - 2 : 0)) {   /* every fragment read of the stage is out */        \
        pre_;                                                                                              \
      }                                                                                                    \
    }                                                                                                      \
  } while (0)
    if (unit_mode) {
#pragma unroll
      for (int t = 0; t < 3; ++t) ROW_TABLE(t, t);
      __builtin_amdgcn_s_barrier();
      GGW_STAMP(wave == 0, 1);
      int ukc = 0, ug = 0;   // this unit's channel chunk and kernel row
      for (int u = 0; u < nfills; ++u) {
        GGW_STAMP(wave == 0 && u < GGW_STAMP_STAGES, 4 + u * 4 + 0);
        const char* abase = smem + (u & 1) * ABUF;
        int kc1 = ukc + 1, g1 = ug;
        if (kc1 == kchunks) {
          kc1 = 0;
          ++g1;
        }
        const int gn = u + 1 < nfills ? g1 : 0;   // (behind the last unit: any valid entries, unused)
        int nsrc[3][TM];
#define ROW_LOAD3 ROW_LOAD(TPG * gn + 0, nsrc[0]); ROW_LOAD(TPG * gn + 1, nsrc[1]); ROW_LOAD(TPG * gn + 2, nsrc[2])
        if (GGW_DIAG(1)) {
          ROW_LOAD3;
        } else {
          STAGE_MFMA(abase, wsm + ((u & 1) * 3 + 0) * WBUF, 0, (void)0);
          STAGE_MFMA(abase, wsm + ((u & 1) * 3 + 1) * WBUF, 1, (void)0);
          STAGE_MFMA(abase, wsm + ((u & 1) * 3 + 2) * WBUF, 2, ROW_LOAD3);   // next unit's rulebook entries under the last k-step
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) ROW_FINISH(nsrc[t], UNIT_SEL(gn, lo0, lo1, lo2), t);   // before the barrier
        ukc = kc1, ug = g1;
        GGW_STAMP(wave == 0 && u < GGW_STAMP_STAGES, 4 + u * 4 + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        GGW_STAMP(wave == 0 && u < GGW_STAMP_STAGES, 4 + u * 4 + 2);
      }
    } else {
      // ---- per-tap stages as one stream of k-steps.  The stage barrier sits at the top of a stage's LAST k-step:
      // by then every fragment read of the stage has landed (its buffers are free for the loaders), and what the
      // barrier publishes -- the next stage's window and weights -- is first needed by the reads that go out behind
      // that k-step's MFMAs.  So the first fragments of stage s+1 land under the last MFMAs of stage s instead of
      // behind a barrier all sixteen waves leave at once (stamps of the barrier-at-the-end loop: ~300 of a stage's
      // 1 800 ticks from the barrier to the first MFMA).  The next stage's row table and fragment addresses are made
      // after the stage's last own read has gone out, in the registers of its own.
      constexpr int NKS = BK / 16;
      static_assert(NKS >= 2, "two k-steps per stage at least");
      ROW_TABLE(0, 0);
      GGW_STAMP(wave == 0, 4 + GGW_STAMP_STAGES * 4 + 3);
      __builtin_amdgcn_s_barrier();          // stage 0's window and weights have landed
      GGW_STAMP(wave == 0, 1);
      unsigned ra[TM], rw[TN];
      int ka[TM], kw[TN];
      bf16x8 fa[2][TM], fw[2][TN];
#define FRAG_SETUP(abase_, wbase_)                                                                         \
  do {                                                                                                     \
    const unsigned a_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(abase_);      \
    const unsigned w_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(wbase_);      \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                       \
      ra[i] = a_lds + (unsigned)aoff[0][i];                                                                \
      ka[i] = akey[0][i] ^ lh;                                                                             \
    }                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                       \
      const int row_ = wn * WTN + j * 32 + lr;                                                             \
      rw[j] = w_lds + (unsigned)(row_ * ROWB);                                                             \
      kw[j] = swz<BK>(row_) ^ lh;                                                                          \
    }                                                                                                      \
  } while (0)
      FRAG_SETUP(smem, wsm);
#pragma unroll
      for (int i = 0; i < TM; ++i) DS_READ_FRAG(fa[0][i], ra[i], ka[i], 0);
#pragma unroll
      for (int j = 0; j < TN; ++j) DS_READ_FRAG(fw[0][j], rw[j], kw[j], 0);
      int st = 0, skc = 0, sg = 0, par = 0, ring = 0;   // this stage's coordinates, window buffer, weight slot
      for (int s = 0; s < nstages; ++s) {
        GGW_STAMP(wave == 0 && s < GGW_STAMP_STAGES, 4 + s * 4 + 0);
        GGW_WSTAMP(s, 2);
#ifdef GGW_KSTAMPS
        kstamp_stage = s;
#endif
        STAGE_NEXT(t1, kc1, g1, st, skc, sg);
        const int ntap = s + 1 < nstages ? TPG * g1 + t1 : 0;   // (behind the last stage: any valid entry, unused -- no branch in the loop)
        const int par1 = (!win || t1 == 0) ? par ^ 1 : par;
        const int ring1 = ring + 1 == RING ? 0 : ring + 1;
        int nsrc[TM];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (ks == NKS - 1) {
            GGW_WSTAMP(s, 0);
            __builtin_amdgcn_s_barrier();
            GGW_WSTAMP(s, 1);
          }
          GGW_KSTAMP(ks);
          // reads of the next k-step: ks + 1 of this stage, or k-step 0 of the next (ra / rw then hold its addresses)
          const int kn = ks + 1 < NKS ? ks + 1 : 0;
          int nxt = 0;
          if (TM * TN <= 2) {
#pragma unroll
            for (; nxt < TM + TN; ++nxt) {
              if (nxt < TM) DS_READ_FRAG(fa[(ks + 1) & 1][nxt], ra[nxt], ka[nxt], kn);
              else DS_READ_FRAG(fw[(ks + 1) & 1][nxt - TM], rw[nxt - TM], kw[nxt - TM], kn);
            }
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(fw[ks & 1][j]), "v"(fa[ks & 1][i]));
#pragma unroll
              for (int r = 0; r < 2; ++r, ++nxt) {
                if (nxt < TM) DS_READ_FRAG(fa[(ks + 1) & 1][nxt], ra[nxt], ka[nxt], kn);
                else if (nxt < TM + TN) DS_READ_FRAG(fw[(ks + 1) & 1][nxt - TM], rw[nxt - TM], kw[nxt - TM], kn);
              }
            }
          if (ks == (NKS >= 3 ? NKS - 3 : 0)) ROW_LOAD(ntap, nsrc);
          if (ks == NKS - 2) {   // the stage's own reads are all out: its address registers take the next stage's
            ROW_FINISH(nsrc, UNIT_SEL(g1, lo0, lo1, lo2), 0);
            FRAG_SETUP(smem + par1 * ABUF, wsm + ring1 * WBUF);
          }
        }
        par = par1, ring = ring1;
        st = t1, skc = kc1, sg = g1;
        GGW_STAMP(wave == 0 && s < GGW_STAMP_STAGES, 4 + s * 4 + 1);
        GGW_STAMP(wave == 0 && s < GGW_STAMP_STAGES, 4 + s * 4 + 2);
      }
      // the reads issued behind the last k-step (unused) have landed.  Their registers are operands of the wait: a
      // result nobody uses would otherwise leave them free for the compiler while the loads are still in flight
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][i])::"memory");
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fw[0][j])::"memory");
    }
  }
  } else {
    // ---- fused waves: a stage's DMA (next stage's weights, and at a fill's first stage the next window) goes out at
    // the top of the stage, lands under its MFMAs and is waited for at its end
    const int wm = wave / NWN, wn = wave % NWN;
    int aoff[3][TM], akey[3][TM];
    const bool stager = wave < NLD;
    if (stager) WIN_FILL(0);
    ROW_TABLE(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    GGW_STAMP(wave == 0, 1);
    constexpr int FNKS = BK / 16;
    static_assert(FNKS >= 2, "two k-steps per stage at least");
    unsigned ra[TM], rw[TN];
    int ka[TM], kw[TN];
    bf16x8 fa[2][TM], fw[2][TN];
    FRAG_SETUP(smem, wsm);
    int st = 0, skc = 0, sg = 0, par = 0, ring = 0, f = 0;   // this stage's coordinates, window buffer, weight slot, fill
    for (int s = 0; s < nstages; ++s) {
      GGW_STAMP(wave == 0 && s < GGW_STAMP_STAGES, 4 + s * 4 + 0);
      // first fragments of the stage: right behind the barrier, ahead of the DMA burst
#pragma unroll
      for (int i = 0; i < TM; ++i) DS_READ_FRAG(fa[0][i], ra[i], ka[i], 0);
#pragma unroll
      for (int j = 0; j < TN; ++j) DS_READ_FRAG(fw[0][j], rw[j], kw[j], 0);
      STAGE_NEXT(t1, kc1, g1, st, skc, sg);
      const bool more = s + 1 < nstages;
      const bool fill_now = (!win || st == 0) && f + 1 < nfills;
      const int ring1 = ring + 1 == RING ? 0 : ring + 1;
      if (stager) {
        if (more) WIN_WLOAD_AT(ring1, g1, kc1, t1);
        if (fill_now) {   // issued last: a window (needed three stages on) stays in flight over this barrier
          if (win) {      // the chunk after this one
            int kcf = skc + 1, gf = sg;
            if (kcf == kchunks) {
              kcf = 0;
              ++gf;
            }
            WIN_FILL_AT(par ^ 1, gf, kcf, 0);
          } else {        // gather mode: the next stage's tap
            WIN_FILL_AT(par ^ 1, g1, kc1, t1);
          }
        }
      }
      GGW_STAMP(wave == 0 && s < GGW_STAMP_STAGES, 4 + s * 4 + 3);   // DMA issued
      int nsrc[TM];
      const int ntap = more ? TPG * g1 + t1 : 0;   // (behind the last stage: any valid entry, unused -- no branch in the loop)
      const int par1 = (!win || t1 == 0) ? par ^ 1 : par;
      // the stage's k-steps; its fragment addresses were made during the previous stage, so the first reads went out
      // right behind the barrier (above), and the next stage's are made once its own last read is out
#pragma unroll
      for (int ks = 0; ks < FNKS; ++ks) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int nxt = 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(fw[ks & 1][j]), "v"(fa[ks & 1][i]));
            if (ks + 1 < FNKS) {
#pragma unroll
              for (int r = 0; r < 2; ++r, ++nxt) {
                if (nxt < TM) DS_READ_FRAG(fa[(ks + 1) & 1][nxt], ra[nxt], ka[nxt], ks + 1);
                else if (nxt < TM + TN) DS_READ_FRAG(fw[(ks + 1) & 1][nxt - TM], rw[nxt - TM], kw[nxt - TM], ks + 1);
              }
            }
          }
        if (ks == (FNKS >= 3 ? FNKS - 3 : 0)) ROW_LOAD(ntap, nsrc);
        if (ks == FNKS - 2) {
          ROW_FINISH(nsrc, UNIT_SEL(g1, lo0, lo1, lo2), 0);
          FRAG_SETUP(smem + par1 * ABUF, wsm + ring1 * WBUF);
        }
      }
      if (!win || t1 == 0) ++f;
      par = par1;
      ring = ring1;
      st = t1, skc = kc1, sg = g1;
      GGW_STAMP(wave == 0 && s < GGW_STAMP_STAGES, 4 + s * 4 + 1);
      GGW_WSTAMP(s, 2);   // (fused loop: slot 2 = MFMAs issued, slot 0 = own DMA landed, slot 1 = barrier passed)
      if (fill_now && win) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      GGW_WSTAMP(s, 0);
      __builtin_amdgcn_s_barrier();
      GGW_WSTAMP(s, 1);
      GGW_STAMP(wave == 0 && s < GGW_STAMP_STAGES, 4 + s * 4 + 2);
    }
  }
  // the MFMAs sit in opaque asm: the hazard checker does not know that the VALU reads of `acc` below follow them
  // (inside the loop no VALU instruction touches the accumulators)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  GGW_STAMP(wave == 0, 2);

  // ---- epilogue.  D row = output channel (reg&3) + 8*(reg>>2) + 4*lh, D col = site lr.  The compute
  // waves park scale*acc+shift as fp32 in the (now free) staging pool; then every wave writes whole
  // output rows -- residual rows are read coalesced here, not 8 bytes per lane from 32 rows.
  TOut* out = reinterpret_cast<TOut*>(a.out);
  constexpr int EROW = BN * 4 + 16;                  // bytes per parked row (+16: rows of a wave on different banks)
  constexpr int PR = BM / EP;                        // rows per epilogue pass
  static_assert((size_t)PR * EROW <= (size_t)POOL, "a pass of the fp32 tile must fit the staging pool");
  static_assert(BM % EP == 0 && PR % WTM == 0, "a compute wave's rows leave in one pass");
  char* et = smem;
  // 16-B stores: 8 channels per lane for bf16 (a store instruction costs the same issue time whatever its width)
  constexpr int CPL = sizeof(TOut) == 2 && BN % 8 == 0 ? 8 : 4;   // channels per lane
  constexpr int LPR = BN / CPL;                    // lanes per output row
  constexpr int RPI = 64 / LPR > 0 ? 64 / LPR : 1; // rows per wave-instruction
  static_assert(LPR <= 64, "row wider than a wavefront");
  const int sub = lane / LPR, cl = (lane % LPR) * CPL;
  f32x4 sc[CPL / 4], sh[CPL / 4];   // this lane's channels are the same for every row it writes
#pragma unroll
  for (int h = 0; h < CPL / 4; ++h) {
    sc[h] = *reinterpret_cast<const f32x4*>(sc_lds + cl + 4 * h);
    sh[h] = *reinterpret_cast<const f32x4*>(sc_lds + BN + cl + 4 * h);
  }
  const bool wide = CPL == 8 && n0 + cl + 8 <= a.cout_total && (a.out_ld & 7) == 0 && (a.res_ld & 7) == 0 &&
                    ((uintptr_t)a.out & 15) == 0 && ((uintptr_t)a.residual & 15) == 0;
  constexpr int NIT = (PR + NW * RPI - 1) / (NW * RPI);   // output rows per lane and pass
#pragma unroll
  for (int ep = 0; ep < EP; ++ep) {
    // the residual rows of the pass are requested before the tile is parked: their round trip runs under the park and
    // its barrier instead of once per output row behind them
    bf16x8 rpre[NIT];
    const bool pre = CPL == 8 && wide && a.residual != nullptr;   // (per lane: `wide` depends on the lane's channels)
    if constexpr (CPL == 8) {
      if (pre) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int rp = wave * RPI + it * NW * RPI + sub, m = m0 + rp + ep * PR;
#pragma unroll
          for (int r = 0; r < 8; ++r) rpre[it][r] = (__bf16)0.f;
          if (rp < PR && m < M) rpre[it] = *reinterpret_cast<const bf16x8*>(a.residual + (size_t)m * a.res_ld + n0 + cl);
        }
      }
    }
    if (!loader) {
      const int wm = wave / NWN, wn = wave % NWN;
      if ((wm * WTM) / PR == ep) {                   // wave-uniform
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int rl = wm * WTM + i * 32 + lr - ep * PR;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int ccl = wn * WTN + j * 32 + 8 * g + 4 * lh;
              f32x4 v;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * g + r];
              *reinterpret_cast<f32x4*>(et + rl * EROW + ccl * 4) = v;   // raw: scale / shift are applied per lane below
            }
          }
        }
      }
    }
    if (ep == 0) GGW_STAMP(wave == 0, 4 + GGW_STAMP_STAGES * 4 + 4);
    __syncthreads();
    if (ep == 0) GGW_STAMP(wave == 0, 4 + GGW_STAMP_STAGES * 4 + 5);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int r0 = wave * RPI + it * NW * RPI;
      const int rp = r0 + sub, rl = rp + ep * PR, m = m0 + rl;
      if (!(rp < PR && m < M)) continue;
      const size_t orow = a.out_rows ? (size_t)a.out_rows[m] : (size_t)m;
#pragma unroll
      for (int h = 0; h < CPL / 4; ++h) {
        const int c = cl + 4 * h;
        f32x4 v = *reinterpret_cast<const f32x4*>(et + rp * EROW + c * 4) * sc[h] + sh[h];
        f32x4 v2 = v;
        if constexpr (CPL == 8) {
          if (wide) {   // both halves, one 16-B residual load and one 16-B store
            if (h == 1) break;
            v2 = *reinterpret_cast<const f32x4*>(et + rp * EROW + c * 4 + 16) * sc[CPL / 4 - 1] + sh[CPL / 4 - 1];
            if (a.residual) {
              const bf16x8 rv = rpre[it];
#pragma unroll
              for (int r = 0; r < 4; ++r) { v[r] += (float)rv[r]; v2[r] += (float)rv[4 + r]; }
            }
            if (a.relu) {
#pragma unroll
              for (int r = 0; r < 4; ++r) { v[r] = fmaxf(v[r], 0.f); v2[r] = fmaxf(v2[r], 0.f); }
            }
            bf16x8 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[r] = (__bf16)v[r]; o[4 + r] = (__bf16)v2[r]; }
            *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(out) + orow * a.out_ld + n0 + c) = o;
            continue;
          }
        }
        if (n0 + c >= a.cout_total) continue;
        if (a.residual) {
          const bf16x4 rv = *reinterpret_cast<const bf16x4*>(a.residual + (size_t)m * a.res_ld + n0 + c);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
        }
        if (a.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        OutOps<TOut>::store4(out + orow * a.out_ld + n0 + c, v);
      }
    }
    if (ep + 1 < EP) __syncthreads();                // the pool is parked into again
  }
  GGW_STAMP(wave == 0, 3);
#endif
}

template <int BM, int BN, int NWM, int NWN, int NWL, int BK, int WMAX, typename TOut, bool UNIT = false, int RING = 3,
          int NTAPS = 9, int EP = 1, bool FUSED = false>
static int launch_win(const GGArgs& a, int groups, hipStream_t stream) {
  constexpr int NT = (NWM * NWN + NWL) * 64;
  constexpr size_t LDS = (size_t)win_pool_bytes<BM, BN, BK, WMAX, UNIT, RING, EP>() + (size_t)NTAPS * BM * 4 + 32 + (size_t)BN * 8;
  static_assert(LDS <= 160 * 1024, "window kernel LDS budget");
  static const int force_gather = getenv("PN_GG_WIN_GATHER") != nullptr;
  auto kern = gg_win_kernel<BM, BN, NWM, NWN, NWL, BK, WMAX, TOut, UNIT, RING, NTAPS, EP, FUSED>;
  static bool attr_set = false;
  if (!attr_set && LDS > 32 * 1024) {
    PN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    attr_set = true;
  }
  // Workgroups go round robin over the eight XCDs by their linear id (x fastest).  With several column tiles or groups
  // the row dimension is padded to a multiple of eight (the extra workgroups exit at once), so that the column tiles
  // of one row tile -- which read the same window -- run on the same XCD and share its L2.
  int gx = pn_div_up(a.rows, BM);
  const int gy = pn_div_up(a.cout_total, BN);
#ifndef GGW_NO_XCD_PAD
  if (gy * groups > 1) gx = (gx + 7) & ~7;
#endif
  const dim3 grid(gx, gy, groups);
  hipLaunchKernelGGL(kern, grid, dim3(NT), LDS, stream, a, force_gather);
  PN_CHECK_LAUNCH("gg_win_kernel");
  return PN_OK;
}

// Tile selection.  Template arguments: <BM, BN, NWM x NWN compute waves, NWL loader waves, BK, WMAX, TOut, UNIT, RING>.
//   Cout <= 64   256 x Cout tiles, unit stages (a kernel row's weight tiles are <= 3 x 8 KB); 128-row tiles when
//                the level has fewer 256-row tiles than CUs
//   Cout <= 128  256 x 128, 4 compute waves of 64 x 128
//   wider        128 x 256 with 8 compute waves of 64 x 64 + 8 loaders (2-6 % ahead of 4 x (64 x 128) + 4 on every
//                256-wide shape); 128 x 64 unit-stage tiles when the level is small (a 90 x 90 map is 64 tiles of
//                128 x 256 on 256 CUs: filling the chip wins 2x although every tile then streams its weights itself)
// PN_GG_WIN_CFG (tuning / tests): P = per-tap stages everywhere, D = 128 x 256 on 4 + 4 waves, T = 128 x 128 tiles
// with a 2-deep weight ring on the 256-wide shapes too (two workgroups per CU; within 3 % of the default there),
// E = 256 x 128 tiles on 256-wide shapes, B = big tiles on small levels, h / t = the two-workgroup narrow configs off.
// Tile policy.  0 (default): a launch is alone on the chip, so small levels get small tiles that fill it.
// 1: the caller keeps several frames in flight on different streams; what counts is CU-time, and a workgroup
// whose LDS pool stays under 80 KB can share a CU with a workgroup of the other frame's kernel -- 128 x 128 tiles
// with the 2-deep ring on the large levels above 64 outputs, and on the small levels the tile with the least
// CU-time instead of the one that fills the chip.  Same stage order and k-chunking as the default tiles, so
// the results are bit-identical under either policy.
static int g_tile_policy = 0;
extern "C" int pn_gg_set_tile_policy(int policy) {
  const int old = g_tile_policy;
  if (policy == 0 || policy == 1) g_tile_policy = policy;
  return old;
}

template <typename TOut>
static int dispatch_win(const GGArgs& a, int groups, hipStream_t stream) {
  const int N = a.cout_total;
  const bool shared_cu = g_tile_policy == 1;
  static const char* force = getenv("PN_GG_WIN_CFG");
  const char f = force ? force[0] : 0;
  const bool unit = f != 'P';
  const long long work_rows = (long long)a.rows * groups;
  if (a.cin % 64 != 0) {   // 32 input channels: 32-channel stages
    if (N <= 32) return unit ? launch_win<256, 32, 4, 1, 2, 32, 352, TOut, true>(a, groups, stream)
                             : launch_win<256, 32, 4, 1, 2, 32, 352, TOut>(a, groups, stream);
    return unit ? launch_win<256, 64, 4, 1, 4, 32, 352, TOut, true>(a, groups, stream)
                : launch_win<256, 64, 4, 1, 4, 32, 352, TOut>(a, groups, stream);
  }
  if (N <= 32) return unit ? launch_win<256, 32, 4, 1, 4, 64, 352, TOut, true>(a, groups, stream)
                           : launch_win<256, 32, 4, 1, 4, 64, 352, TOut>(a, groups, stream);
  if (N <= 64) {
    // 64 -> 64 on a large level: 32-channel units keep the pool at 79 KB, so two workgroups share a CU and one's
    // prologue / epilogue (half of a tile's time on this level) runs under the other's units: 42.0 -> 34.8 us on
    // the 245 k-row level.  Wider inputs (256 -> 64) lose more to the doubled unit count than they gain.  'h' = off.
    // 32-channel chunks change the summation order, so only launches that declare `free_order` (the sparse
    // backbone) take it: the ranked head must reproduce the dense head's bits from another launch shape.
    if (unit && a.cin == 64 && a.free_order && f != 'h') return launch_win<256, 64, 4, 1, 4, 32, 320, TOut, true>(a, groups, stream);
    // (sharing the chip: the 256-row tiles below take half the workgroups for the same launch time, +1 % on the frame)
    if (unit && work_rows <= 256 * 200 && f != 'B' && !shared_cu) return launch_win<128, 64, 2, 1, 2, 64, 160, TOut, true>(a, groups, stream);
    return unit ? launch_win<256, 64, 4, 1, 4, 64, 352, TOut, true>(a, groups, stream)
                : launch_win<256, 64, 4, 1, 4, 64, 352, TOut>(a, groups, stream);
  }
  // <= 128 outputs: 128 x 128 tiles with a 2-deep weight ring (77 KB, two workgroups per CU): 53.0 -> 50.6 us on the
  // 111 k-row 128 -> 128 level against 256 x 128 tiles that own a CU.  't' = off.
  const bool small_level = work_rows <= 128 * 160 && N % 128 == 0;
  const long long in_rows = a.in_ld > 0 ? (long long)a.in_bytes / (2ll * a.in_ld) : 0;
  const bool same_grid = in_rows <= (long long)a.rows + a.rows / 4;     // stride 1: as many input rows as output sites
  // sharing the chip: 256 x 256 tiles on fused waves take half the CUs for 1.5x the launch time of the 128 x 256
  // tiles (a quarter less CU-time); the other frame's kernels run on the rest ('f' = off, 'F' = on without the policy)
  if ((f == 'F' || (shared_cu && f != 'f')) && N > 128 && !small_level && same_grid)
    return launch_win<256, 256, 4, 2, 0, 64, 264, TOut, false, 2, 9, 2, true>(a, groups, stream);
  if (f == 'T' || ((N <= 128 || (shared_cu && !small_level)) && f != 't')) return launch_win<128, 128, 2, 2, 4, 64, 144, TOut, false, 2>(a, groups, stream);
  if (N <= 128) return launch_win<256, 128, 4, 1, 4, 64, 352, TOut>(a, groups, stream);
  if (work_rows <= 128 * 160 && N % 128 == 0 && f != 'B') {   // (WMAX 160: a stride-2 run is 2 * 64 + 3 rows)
    // 128 x 64 tiles (4 compute waves of 32 x 64 + 4 loaders): as many workgroups as 64 x 128 tiles give, but a
    // workgroup streams half the weight bytes for the same MFMAs (the level is bound by DMA issue): +2 % on the
    // frame.  (WMAX 288: a stride-2 run is 2 * 128 + 3 rows.)  'Q' keeps the 64 x 128 shape for comparison.
    // sharing the chip with another frame: 256 x 64 tiles -- 128 workgroups on half the CUs, 22 us instead of 19.6,
    // but a third less CU-time than 256 tiles of 128 x 64 (a tile streams the same weights for twice the rows):
    // +3 % on the frame with two lanes ('c' = off, 'C' = on without the policy)
    if (f == 'C' || (shared_cu && f != 'c')) return launch_win<256, 64, 4, 1, 4, 64, 352, TOut, true>(a, groups, stream);
    if (f == 'Q') return launch_win<64, 128, 2, 2, 4, 64, 160, TOut, true>(a, groups, stream);
    return launch_win<128, 64, 4, 1, 4, 64, 288, TOut, true>(a, groups, stream);
  }
  if (f == 'D') return launch_win<128, 256, 2, 2, 4, 64, 192, TOut>(a, groups, stream);
  if (f == 'E') return launch_win<256, 128, 4, 1, 4, 64, 352, TOut>(a, groups, stream);
  return launch_win<128, 256, 2, 4, 8, 64, 192, TOut>(a, groups, stream);
}

// 1-tap GEMMs (the head's per-tap products, the parity GEMMs of the transposed convs): the same pipeline with the
// tile's own rows as its window.  Narrow outputs stay on the per-tap kernel.
template <typename TOut>
static int dispatch_win1(const GGArgs& a, int groups, hipStream_t stream) {
  if (a.cin % 64 != 0 || a.cout_total <= 64) return -1;
  if (a.cout_total <= 128) return launch_win<256, 128, 4, 1, 4, 64, 352, TOut, false, 3, 1>(a, groups, stream);
  return launch_win<128, 256, 2, 4, 8, 64, 192, TOut, false, 3, 1>(a, groups, stream);
}

int pn_gg_try_window(const GGArgs& a, int groups, bool out_fp32, hipStream_t stream) {
  static const bool off = getenv("PN_GG_NO_WIN") != nullptr;
  static const bool off1 = getenv("PN_GG_NO_WIN1") != nullptr;
  if (off || a.in_bytes == 0 || a.diag == 3) return -1;
  if (a.ntaps == 1 && !off1)
    return out_fp32 ? dispatch_win1<float>(a, groups, stream) : dispatch_win1<__bf16>(a, groups, stream);
  if (a.ntaps != 9 || a.nbr == nullptr) return -1;
  return out_fp32 ? dispatch_win<float>(a, groups, stream) : dispatch_win<__bf16>(a, groups, stream);
}
