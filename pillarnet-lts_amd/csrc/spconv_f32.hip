// fp32 rulebook convolution for the training path (forward and, with transposed weights and the inverted rulebook,
// the data gradient), gfx950:
//   out[m][co] = act(scale[co] * sum_{tap, ci} in[nbr[tap][m]][ci] * W[tap][co][ci] + shift[co] (+ residual[m][co]))
// The reference trains in fp32 and leaves this product to spconv (third-party; its call sites:
// det3d/models/backbones/base.py:9-10,139-140, PillarResNet.py:73-150), which gathers per tap and calls a GEMM.
//
// One workgroup owns 64 * NS output sites x ALL output channels, one wavefront per 64 sites x 64 channels
// (v_mfma_f32_32x32x2_f32, 2 x 2 accumulator tiles).  A stage is (tap, 16 input channels): the 16-channel slices of the
// tap's weight rows and of the sites' gathered input rows are fetched once per workgroup as 16-byte pieces, staged
// through LDS (double-buffered: stage k + 1 is in flight to registers while stage k multiplies; one barrier per
// stage) and read back in the MFMA operand layout -- lane l holds row l % 32 at reduction index l / 32.  Rows sit 17
// floats apart in LDS so that the 32 rows a half wave reads fall on different banks.  The per-wave kernel this
// replaces for wide layers (spconv.hip) made every wavefront stream the whole 9 x cout x cin weight tensor for its
// 32 sites: 8.7 GB through L2 per launch on the 256-wide level, 72 TFLOP/s.
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#ifndef SF_KS_DEF
#define SF_KS_DEF 16
#endif
static constexpr int SF_KS = SF_KS_DEF;   // input channels per stage
static constexpr int SF_PPR = SF_KS / 4;  // 16-byte pieces per staged row
static constexpr int SF_LD = SF_KS + 1;   // LDS row stride in floats

struct SpconvF32Args {
  const float* in;        // [in_rows, cin]
  const int32_t* nbr;     // [9][nbr_stride]
  const int32_t* count;   // device row count
  const float* wt;        // [9][cout][cin]
  const float* scale;     // [cout]
  const float* shift;     // [cout]
  const float* residual;  // [cap, cout] or NULL
  float* out;             // [cap, cout]
  int nbr_stride, cap, cin, cout, relu;
};

template <int NCO, int NS>
__global__ __launch_bounds__(64 * NCO * NS) void spconv_f32_lds_kernel(SpconvF32Args a) {
  constexpr int NT = 64 * NCO * NS, COUT = 64 * NCO, TS = 64 * NS;
  constexpr int NW = COUT * SF_PPR / NT, NI = TS * SF_PPR / NT;   // 16-byte pieces per thread and stage: weight rows, input rows
  static_assert(COUT * SF_PPR % NT == 0 && TS * SF_PPR % NT == 0, "a stage's rows must divide over the threads");
  constexpr int BUF = (COUT + TS) * SF_LD;               // floats per stage buffer: weight rows, then site rows
  extern __shared__ __attribute__((aligned(16))) float sf_lds[];   // [2][BUF]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int cw = (wave % NCO) * 64, sw = (wave / NCO) * 64;       // this wavefront's channels / sites inside the tile
  int M = *a.count;
  if (M > a.cap) M = a.cap;
  const int m0 = blockIdx.x * TS;
  if (m0 >= M) return;                                             // workgroup-uniform
  const int kchunks = a.cin / SF_KS, nstages = 9 * kchunks;

  f32x16 acc[2][2];                                                // [channel tile][site tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // this thread's pieces: weight row / input site q / SF_PPR, 16-byte part q % SF_PPR of the row's slice
  f32x4 rw[NW], ri[NI];
  int ix[NI], ixn[NI];                                             // rulebook entries of the current / the next tap
#define SF_IDX(dst_, tap_)                                                         \
  _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                 \
    const int m = m0 + (tid + j * NT) / SF_PPR;                                         \
    dst_[j] = (m < M && (tap_) < 9) ? a.nbr[(size_t)(tap_) * a.nbr_stride + m] : -1; \
  }
#define SF_LOAD(tap_, kc_)                                                                                     \
  _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                                             \
    const int q = tid + j * NT;                                                                                \
    rw[j] = *reinterpret_cast<const f32x4*>(a.wt + ((size_t)(tap_) * COUT + q / SF_PPR) * a.cin + (kc_) * SF_KS + (q % SF_PPR) * 4); \
  }                                                                                                            \
  _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                                             \
    const int q = tid + j * NT;                                                                                \
    ri[j] = ix[j] >= 0 ? *reinterpret_cast<const f32x4*>(a.in + (size_t)ix[j] * a.cin + (kc_) * SF_KS + (q % SF_PPR) * 4) \
                       : f32x4{0.f, 0.f, 0.f, 0.f};                                                            \
  }
#define SF_STORE(buf_)                                                                       \
  _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                           \
    const int q = tid + j * NT;                                                              \
    float* d = sf_lds + (buf_) * BUF + (q / SF_PPR) * SF_LD + (q % SF_PPR) * 4;                        \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) d[e] = rw[j][e];                           \
  }                                                                                          \
  _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                           \
    const int q = tid + j * NT;                                                              \
    float* d = sf_lds + (buf_) * BUF + (COUT + q / SF_PPR) * SF_LD + (q % SF_PPR) * 4;                 \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) d[e] = ri[j][e];                           \
  }
  SF_IDX(ix, 0);
  SF_IDX(ixn, 1);
  SF_LOAD(0, 0);
  SF_STORE(0);
  __syncthreads();
  int tap = 0, kc = 0, buf = 0;
  for (int s = 0; s < nstages; ++s, buf ^= 1) {
    int tap1 = tap, kc1 = kc + 1;                                  // the next stage
    if (kc1 == kchunks) {
      kc1 = 0;
      ++tap1;
#pragma unroll
      for (int j = 0; j < NI; ++j) ix[j] = ixn[j];
      SF_IDX(ixn, tap1 + 1);                                       // (entries of the tap after: a tap ahead of their use)
    }
    const bool more = s + 1 < nstages;
    if (more) {
      SF_LOAD(tap1, kc1);
    }
    const float* wl = sf_lds + buf * BUF + (cw + lr) * SF_LD + lh;
    const float* il = sf_lds + buf * BUF + (COUT + sw + lr) * SF_LD + lh;
#pragma unroll
    for (int ks = 0; ks < SF_KS / 2; ++ks) {
      const float w0 = wl[2 * ks], w1 = wl[32 * SF_LD + 2 * ks], x0 = il[2 * ks], x1 = il[32 * SF_LD + 2 * ks];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, x0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, x1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, x0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, x1, acc[1][1], 0, 0, 0);
    }
    if (more) {
      SF_STORE(buf ^ 1);
    }
    __syncthreads();
    tap = tap1, kc = kc1;
  }
#undef SF_IDX
#undef SF_LOAD
#undef SF_STORE

  // epilogue: D row = channel (r & 3) + 8 * (r >> 2) + 4 * lh of the tile, D col = site lr: four consecutive channels of
  // one site per register group
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + sw + 32 * j + lr;
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = cw + 32 * i + 8 * g + 4 * lh;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + c0);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + c0);
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * g + r];
        v = v * sc + sh;
        if (a.residual) v += *reinterpret_cast<const f32x4*>(a.residual + (size_t)m * COUT + c0);
        if (a.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        *reinterpret_cast<f32x4*>(a.out + (size_t)m * COUT + c0) = v;
      }
  }
}

template <int NCO, int NS>
static int launch_f32_lds(const SpconvF32Args& a, hipStream_t stream) {
  constexpr int TS = 64 * NS;
  constexpr size_t lds = 2ull * (64 * NCO + TS) * SF_LD * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    PN_CHECK_HIP(hipFuncSetAttribute((const void*)spconv_f32_lds_kernel<NCO, NS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL((spconv_f32_lds_kernel<NCO, NS>), dim3((unsigned)pn_div_up(a.cap, TS)), dim3(64 * NCO * NS), lds,
                     stream, a);
  PN_CHECK_LAUNCH("spconv_f32_lds_kernel");
  return PN_OK;
}

// -1: not a shape of this kernel (the caller falls back to the per-wave kernel)
int pn_spconv_f32_lds(const float* in, int cin, const int32_t* nbr, int nbr_stride, const int32_t* count_dev, int cap,
                      const float* weight, const float* scale, const float* shift, const float* residual, int relu,
                      float* out, int cout, hipStream_t stream) {
  if (cin % SF_KS != 0 || (cout != 64 && cout != 128 && cout != 256)) return -1;
  SpconvF32Args a;
  a.in = in; a.nbr = nbr; a.count = count_dev; a.wt = weight; a.scale = scale; a.shift = shift;
  a.residual = residual; a.out = out; a.nbr_stride = nbr_stride; a.cap = cap; a.cin = cin; a.cout = cout; a.relu = relu;
  if (cout == 256) return launch_f32_lds<4, 4>(a, stream);
  if (cout == 128) return launch_f32_lds<2, 4>(a, stream);
  return launch_f32_lds<1, 4>(a, stream);
}
