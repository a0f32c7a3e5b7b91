// Weight gradient of a rulebook convolution in fp32 (training runs in fp32, as the reference does):
//   dW[co][tap][ci] = sum_m G[m][co] * X[nbr[tap][m]][ci]
// i.e. per tap a [cout x sites] . [sites x cin] product whose long dimension is the site list.  spconv's own
// backward (the reference calls it through spconv.pytorch, a third-party package) does the same per tap with a
// gather + GEMM; here the gather is the B-operand load itself.
//
// Shape of the work: the output is small (<= 9 x 256 x 256) and the reduction long (1e5 .. 1e6 sites), so the site
// list is cut into chunks and the per-chunk partial sums are added up by a second kernel in a fixed order
// (bit-reproducible: no float atomics).  The MFMA operand layout of v_mfma_f32_32x32x2_f32 -- lane l holds row l % 32
// of site l / 32 -- is a 128-B run of one gradient row and of one gathered input row per half wave: no transposes.
// Two kernels: widths that are multiples of 64 go through `wgrad_lds_kernel` (one workgroup per (chunk, tap) owns the
// whole cout x cin output and stages whole rows through LDS: 93-108 TFLOP/s on the 128- and 256-wide levels against
// 56 for the per-tile kernel); 32-wide layers stay on `wgrad_kernel`, one wavefront per (chunk, tap, output tile)
// reading its operands straight from global memory.
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct WgradArgs {
  const float* x;      // [n_in, x_ld]
  const float* g;      // [rows, g_ld]
  const int32_t* nbr;  // [ntaps][nbr_stride] input row of (tap, output site) or -1; NULL: identity (ntaps == 1)
  const int32_t* count;
  float* partial;      // [chunks][ntaps][cout][cin]
  int x_ld, g_ld, nbr_stride, rows, cin, cout, ntaps, chunk_sites;
};

template <int TCO, int TCI, int U>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
  const int tiles_ci = a.cin / (32 * TCI), ntiles = tiles_ci * (a.cout / (32 * TCO));
  const int tile = blockIdx.z * (blockDim.x >> 6) + wave;
  if (tile >= ntiles) return;                       // wave-uniform
  const int co0 = (tile / tiles_ci) * 32 * TCO, ci0 = (tile % tiles_ci) * 32 * TCI;
  const int tap = blockIdx.y, chunk = blockIdx.x;
  int M = a.count ? *a.count : a.rows;
  if (M > a.rows) M = a.rows;
  const int s_begin = chunk * a.chunk_sites;
  const int s_end = min(M, s_begin + a.chunk_sites);
  f32x16 acc[TCO][TCI];
#pragma unroll
  for (int i = 0; i < TCO; ++i)
#pragma unroll
    for (int j = 0; j < TCI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int32_t* nbr = a.nbr ? a.nbr + (size_t)tap * a.nbr_stride : nullptr;
  for (int s0 = s_begin; s0 < s_end; s0 += 2 * U) {
    float ga[U][TCO], xb[U][TCI];
    int idx[U];
#pragma unroll
    for (int p = 0; p < U; ++p) {                   // two sites per MFMA: the half wave picks one
      const int site = s0 + 2 * p + lh;
      idx[p] = site < s_end ? (nbr ? nbr[site] : site) : -1;
    }
#pragma unroll
    for (int p = 0; p < U; ++p) {
      const int site = s0 + 2 * p + lh;
      const bool ok = site < s_end;
      const float* gr = a.g + (size_t)(ok ? site : 0) * a.g_ld + co0 + lr;
      const float* xr = a.x + (size_t)(idx[p] >= 0 ? idx[p] : 0) * a.x_ld + ci0 + lr;
#pragma unroll
      for (int t = 0; t < TCO; ++t) ga[p][t] = ok ? gr[32 * t] : 0.f;
#pragma unroll
      for (int t = 0; t < TCI; ++t) xb[p][t] = idx[p] >= 0 ? xr[32 * t] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < U; ++p)
#pragma unroll
      for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TCI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[p][i], xb[p][j], acc[i][j], 0, 0, 0);
  }
  // D[i][j]: lane holds column j = lr (input channel) and rows i = (r & 3) + 8 * (r >> 2) + 4 * lh (output channel)
  float* P = a.partial + ((size_t)chunk * a.ntaps + tap) * (size_t)a.cout * a.cin;
#pragma unroll
  for (int i = 0; i < TCO; ++i)
#pragma unroll
    for (int j = 0; j < TCI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        P[(size_t)co * a.cin + ci0 + 32 * j + lr] = acc[i][j][r];
      }
}

// The same product with the operand rows staged through LDS, for widths that are multiples of 64: one workgroup owns the
// WHOLE cout x cin output of its (chunk, tap) -- one wavefront per 64 x 64 tile, 16 of them on a 256 x 256 layer -- and
// every gradient row and every gathered input row of the chunk is fetched from memory once per workgroup, as whole rows
// (16 bytes per lane), instead of once per 64-wide tile (the per-tile kernel above asks L2 for 7.9 GB per launch on the
// 256-wide level, 44 % of it misses, and its MFMA pipe is 39 % busy).  Groups of S sites are double-buffered: while
// the MFMAs of group k read buffer k & 1, the rows of group k + 1 are in flight to registers and the rulebook entries
// of group k + 2 behind them; one barrier per group.
typedef __attribute__((ext_vector_type(4))) float wg_f32x4;
template <int NCO, int NCI, int S>
__global__ __launch_bounds__(64 * NCO * NCI) void wgrad_lds_kernel(WgradArgs a) {
  constexpr int NT = 64 * NCO * NCI, COUT = 64 * NCO, CIN = 64 * NCI;
  constexpr int ROWF = COUT + CIN + 32;            // floats per staged site: gradient row, input row, +32 so that the two
                                                   // sites a wavefront reads at once sit on different banks
  constexpr int GQ = COUT / 4, XQ = CIN / 4;       // 16-B pieces per row
  constexpr int NG = S * GQ / NT, NX = S * XQ / NT;
  static_assert(S * GQ % NT == 0 && S * XQ % NT == 0 && S % 2 == 0, "a group's rows must divide over the threads");
  extern __shared__ __attribute__((aligned(16))) float wg_lds[];   // [2][S][ROWF]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int co0 = (wave / NCI) * 64, ci0 = (wave % NCI) * 64;
  const int tap = blockIdx.y, chunk = blockIdx.x;
  int M = a.count ? *a.count : a.rows;
  if (M > a.rows) M = a.rows;
  const int s_begin = chunk * a.chunk_sites;
  const int s_end = min(M, s_begin + a.chunk_sites);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int32_t* nbr = a.nbr ? a.nbr + (size_t)tap * a.nbr_stride : nullptr;
  wg_f32x4 rg[NG], rx[NX];
  int ix[NX];
#define WL_IDX(g0_)                                                    \
  _Pragma("unroll") for (int j = 0; j < NX; ++j) {                     \
    const int site = (g0_) + (tid + j * NT) / XQ;                      \
    ix[j] = site < s_end ? (nbr ? nbr[site] : site) : -1;              \
  }
#define WL_ROWS(g0_)                                                                                         \
  _Pragma("unroll") for (int j = 0; j < NG; ++j) {                                                           \
    const int q = tid + j * NT, site = (g0_) + q / GQ;                                                       \
    rg[j] = site < s_end ? *reinterpret_cast<const wg_f32x4*>(a.g + (size_t)site * a.g_ld + (q % GQ) * 4)    \
                         : wg_f32x4{0.f, 0.f, 0.f, 0.f};                                                     \
  }                                                                                                          \
  _Pragma("unroll") for (int j = 0; j < NX; ++j) {                                                           \
    const int q = tid + j * NT;                                                                              \
    rx[j] = ix[j] >= 0 ? *reinterpret_cast<const wg_f32x4*>(a.x + (size_t)ix[j] * a.x_ld + (q % XQ) * 4)     \
                       : wg_f32x4{0.f, 0.f, 0.f, 0.f};                                                       \
  }
#define WL_STORE(buf_)                                                                                       \
  _Pragma("unroll") for (int j = 0; j < NG; ++j) {                                                           \
    const int q = tid + j * NT;                                                                              \
    *reinterpret_cast<wg_f32x4*>(wg_lds + ((buf_) * S + q / GQ) * ROWF + (q % GQ) * 4) = rg[j];              \
  }                                                                                                          \
  _Pragma("unroll") for (int j = 0; j < NX; ++j) {                                                           \
    const int q = tid + j * NT;                                                                              \
    *reinterpret_cast<wg_f32x4*>(wg_lds + ((buf_) * S + q / XQ) * ROWF + COUT + (q % XQ) * 4) = rx[j];       \
  }
  WL_IDX(s_begin);
  WL_ROWS(s_begin);
  WL_IDX(s_begin + S);
  WL_STORE(0);
  __syncthreads();
  int buf = 0;
  for (int g0 = s_begin; g0 < s_end; g0 += S, buf ^= 1) {
    WL_ROWS(g0 + S);                               // (past s_end: zeros, never used)
    WL_IDX(g0 + 2 * S);
    const float* base = wg_lds + (buf * S + lh) * ROWF + lr;   // two sites per MFMA: the half wave picks one
#pragma unroll
    for (int p = 0; p < S / 2; ++p) {
      const float* r = base + 2 * p * ROWF;
      const float g0v = r[co0], g1v = r[co0 + 32], x0v = r[COUT + ci0], x1v = r[COUT + ci0 + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0v, x0v, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0v, x1v, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1v, x0v, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1v, x1v, acc[1][1], 0, 0, 0);
    }
    WL_STORE(buf ^ 1);
    __syncthreads();
  }
#undef WL_IDX
#undef WL_ROWS
#undef WL_STORE
  // D[i][j]: lane holds column j = lr (input channel) and rows i = (r & 3) + 8 * (r >> 2) + 4 * lh (output channel)
  float* P = a.partial + ((size_t)chunk * a.ntaps + tap) * (size_t)a.cout * a.cin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        P[(size_t)co * a.cin + ci0 + 32 * j + lr] = acc[i][j][r];
      }
}

// dw[co][tap][ci] = sum over chunks, in chunk order per lane group and a fixed tree over the 8 groups
__global__ __launch_bounds__(512) void wgrad_reduce_kernel(const float* partial, int chunks, int ntaps, int cout, int cin,
                                                           float* dw) {
  __shared__ float part[8][64];
  const int o = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  const size_t n_out = (size_t)ntaps * cout * cin;
  float s = 0.f;
  if ((size_t)o < n_out)
    for (int c = grp; c < chunks; c += 8) s += partial[(size_t)c * n_out + o];
  part[grp][threadIdx.x & 63] = s;
  __syncthreads();
  if (grp == 0 && (size_t)o < n_out) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += part[k][threadIdx.x];
    const int ci = o % cin, co = (o / cin) % cout, tap = o / (cin * cout);
    dw[((size_t)co * ntaps + tap) * cin + ci] = t;
  }
}

#ifndef WG_S_DEF
#define WG_S_DEF 16
#endif
static constexpr int WG_S = WG_S_DEF;   // sites per staged group of the LDS kernel

// widths the LDS-staged kernel is instantiated for (one wavefront per 64 x 64 tile, at most 16)
static bool wgrad_lds_shape(int cin, int cout) {
  if (cin % 64 || cout % 64) return false;
  const int nco = cout / 64, nci = cin / 64;
  return (nco == 4 && (nci == 4 || nci == 2)) || (nco == 2 && (nci == 4 || nci == 2 || nci == 1)) ||
         (nco == 1 && (nci == 2 || nci == 1));
}

static int wgrad_plan(int rows, int cin, int cout, int ntaps, int* chunk_sites) {
  if (wgrad_lds_shape(cin, cout)) {
    // one workgroup per (chunk, tap): about two workgroups per CU's worth of wavefront slots, partial sums within 80 MiB
    const int waves = (cin / 64) * (cout / 64);
    long long chunks = (256ll * 16 / waves) / ntaps;
    const long long by_mem = (80ll << 20) / ((long long)ntaps * cout * cin * 4);
    if (chunks > by_mem) chunks = by_mem;
    const long long by_rows = (rows + 8 * WG_S - 1) / (8 * WG_S);
    if (chunks > by_rows) chunks = by_rows;
    if (chunks < 1) chunks = 1;
    int per = (int)((rows + chunks - 1) / chunks);
    per = (per + WG_S - 1) / WG_S * WG_S;
    if (per < WG_S) per = WG_S;
    *chunk_sites = per;
    return (int)((rows + per - 1) / per > 0 ? (rows + per - 1) / per : 1);
  }
  // enough wavefronts to fill the chip a few times over, partial sums within 64 MiB
  const int tco = cout % 64 == 0 ? 2 : 1, tci = cin % 64 == 0 ? 2 : 1;
  const long long ntiles = (long long)(cout / (32 * tco)) * (cin / (32 * tci));
  long long chunks = 8192 / (ntiles * ntaps);
  const long long by_mem = (64ll << 20) / ((long long)ntaps * cout * cin * 4);
  if (chunks > by_mem) chunks = by_mem;
  const long long by_rows = (rows + 255) / 256;
  if (chunks > by_rows) chunks = by_rows;
  if (chunks < 1) chunks = 1;
  int per = (int)((rows + chunks - 1) / chunks);
  per = (per + 15) / 16 * 16;
  if (per < 16) per = 16;
  *chunk_sites = per;
  return (int)((rows + per - 1) / per > 0 ? (rows + per - 1) / per : 1);
}

extern "C" size_t pn_spconv_wgrad_workspace_bytes(int rows, int cin, int cout, int ntaps) {
  if (rows < 0 || cin <= 0 || cout <= 0 || ntaps <= 0) return 0;
  int per;
  const int chunks = wgrad_plan(rows, cin, cout, ntaps, &per);
  return (size_t)chunks * ntaps * cout * cin * sizeof(float);
}

extern "C" int pn_spconv_wgrad(const float* x, int cin, int x_ld, const int32_t* nbr, int nbr_stride, int ntaps,
                               const int32_t* count_dev, int rows, const float* grad_out, int cout, int g_ld,
                               float* dw, void* workspace, size_t workspace_bytes, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(x && grad_out && dw && rows >= 0 && ntaps >= 1 && ntaps <= 9,
               "pn_spconv_wgrad: bad argument (null pointer, rows %d, taps %d)", rows, ntaps);
  PN_CHECK_ARG(cin > 0 && cout > 0 && cin % 32 == 0 && cout % 32 == 0,
               "pn_spconv_wgrad: channel widths %d -> %d must be multiples of 32", cin, cout);
  PN_CHECK_ARG(nbr || ntaps == 1, "pn_spconv_wgrad: %d taps need a rulebook", ntaps);
  if (x_ld == 0) x_ld = cin;
  if (g_ld == 0) g_ld = cout;
  PN_CHECK_ARG(x_ld >= cin && g_ld >= cout && (!nbr || nbr_stride >= rows),
               "pn_spconv_wgrad: row strides (%d, %d) / rulebook stride %d smaller than the data", x_ld, g_ld, nbr_stride);
  int per;
  const int chunks = wgrad_plan(rows, cin, cout, ntaps, &per);
  const size_t need = (size_t)chunks * ntaps * cout * cin * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    pn_set_error("pn_spconv_wgrad: workspace of %zu bytes, %zu needed (pn_spconv_wgrad_workspace_bytes)", workspace_bytes, need);
    return PN_ENOSPC;
  }
  WgradArgs a;
  a.x = x; a.g = grad_out; a.nbr = nbr; a.count = count_dev; a.partial = (float*)workspace;
  a.x_ld = x_ld; a.g_ld = g_ld; a.nbr_stride = nbr_stride; a.rows = rows; a.cin = cin; a.cout = cout;
  a.ntaps = ntaps; a.chunk_sites = per;
  const size_t n_out = (size_t)ntaps * cout * cin;
  // (the LDS kernel fetches whole rows 16 bytes per lane; rows that are not 16-byte aligned -- odd strides of a view --
  // take the per-tile kernel on the same chunks)
  if (wgrad_lds_shape(cin, cout) && x_ld % 4 == 0 && g_ld % 4 == 0 && (((uintptr_t)x | (uintptr_t)grad_out) & 15) == 0) {
    const dim3 grid(chunks, ntaps);
#define WG_LDS_CASE(NCO_, NCI_)                                                                              \
  if (cout == 64 * NCO_ && cin == 64 * NCI_) {                                                               \
    constexpr size_t lds = 2ull * WG_S * (64 * NCO_ + 64 * NCI_ + 32) * sizeof(float);                       \
    static bool attr_done = false;                                                                           \
    if (!attr_done) {                                                                                        \
      PN_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_lds_kernel<NCO_, NCI_, WG_S>,                      \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
      attr_done = true;                                                                                      \
    }                                                                                                        \
    hipLaunchKernelGGL((wgrad_lds_kernel<NCO_, NCI_, WG_S>), grid, dim3(64 * NCO_ * NCI_), lds, stream, a);  \
  } else
    WG_LDS_CASE(4, 4) WG_LDS_CASE(4, 2) WG_LDS_CASE(2, 4) WG_LDS_CASE(2, 2) WG_LDS_CASE(2, 1) WG_LDS_CASE(1, 2) WG_LDS_CASE(1, 1)
    { pn_set_error("pn_spconv_wgrad: no LDS kernel for %d -> %d", cin, cout); return PN_EINVAL; }
#undef WG_LDS_CASE
    PN_CHECK_LAUNCH("wgrad_lds_kernel");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n_out + 63) / 64)), dim3(512), 0, stream,
                       (const float*)workspace, chunks, ntaps, cout, cin, dw);
    PN_CHECK_LAUNCH("wgrad_reduce_kernel");
    return PN_OK;
  }
  const int tco = cout % 64 == 0 ? 2 : 1, tci = cin % 64 == 0 ? 2 : 1;
  const int ntiles = (cout / (32 * tco)) * (cin / (32 * tci));
  const int wpb = ntiles < 4 ? ntiles : 4;                 // wavefronts (tiles) per workgroup
  const dim3 grid(chunks, ntaps, (ntiles + wpb - 1) / wpb), block(64 * wpb);
  if (tco == 2 && tci == 2) hipLaunchKernelGGL((wgrad_kernel<2, 2, 4>), grid, block, 0, stream, a);
  else if (tco == 2) hipLaunchKernelGGL((wgrad_kernel<2, 1, 4>), grid, block, 0, stream, a);
  else if (tci == 2) hipLaunchKernelGGL((wgrad_kernel<1, 2, 4>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((wgrad_kernel<1, 1, 8>), grid, block, 0, stream, a);
  PN_CHECK_LAUNCH("wgrad_kernel");
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n_out + 63) / 64)), dim3(512), 0, stream,
                     (const float*)workspace, chunks, ntaps, cout, cin, dw);
  PN_CHECK_LAUNCH("wgrad_reduce_kernel");
  return PN_OK;
}
