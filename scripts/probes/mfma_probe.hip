// Micro-probe: one wave per SIMD issuing v_mfma_f32_32x32x16_bf16 with R ds_read_b128 and V VALU ops per MFMA.
// Prints shader cycles (s_memtime) and wall time per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NWAVES, int READS8, int VALU8>   // reads / valu ops per group of 8 MFMAs
__global__ __launch_bounds__(NWAVES * 64) void probe(int iters, float* sink, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += NWAVES * 64) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 127);
  __syncthreads();
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 f[8];
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned addr = base + (unsigned)(((lane & 31) * 128) + ((((lane >> 5)) ^ ((lane >> 1) & 7)) << 4));
  for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(f[i]) : "v"(addr + i * 4096));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned v0 = lane, v1 = tid;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(f[m]), "v"(f[(m + 1) & 7]));
      if (m < READS8) asm volatile("ds_read_b128 %0, %1" : "=v"(f[(m + 4) & 7]) : "v"(addr + ((it + m) & 7) * 4096));
#pragma unroll
      for (int v = 0; v < (VALU8 + 7 - m) / 8; ++v) asm volatile("v_xad_u32 %0, %0, 32, %1" : "+v"(v0) : "v"(v1));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 8; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[i])::"memory");
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + (float)f[i][0];
  if (s == 12345.f) sink[0] = s + v0;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}

template <int NWAVES, int READS8, int VALU8>
void run(float* sink, unsigned long long* cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NWAVES, READS8, VALU8>), dim3(256), dim3(NWAVES * 64), 65536, 0, iters, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("waves/CU %d  ds_read_b128 per 8 MFMA %d  VALU per 8 MFMA %2d : %5.1f ticks/MFMA  %5.2f ns/MFMA/wave  (tick rate %.2f GHz, %.0f TF/s chip)\n",
         NWAVES, READS8, VALU8, (double)c / (iters * 8.0), ms * 1e6 / (iters * 8.0), c / (ms * 1e6),
         256.0 * NWAVES * iters * 8 * 32768.0 / (ms * 1e-3) / 1e12);
}

// the window kernel's k-step: G MFMAs on G accumulators, R reads requested at the top, one wait per k-step
template <int NWAVES, int G, int R, int RANDOM = 0>
__global__ __launch_bounds__(NWAVES * 64) void kstep(int iters, float* sink, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += NWAVES * 64) {
    if (RANDOM) {   // two random bf16 per dword, magnitudes in [2^-8, 2^-7): the operands' bits toggle like real data
      unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      const unsigned lo = (h & 0x807fu) | 0x3b80u, hi = ((h >> 16) & 0x807fu) | 0x3b80u;
      reinterpret_cast<unsigned*>(smem)[i] = lo | (hi << 16);
    } else {
      reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 127);
    }
  }
  __syncthreads();
  f32x16 acc[G];
  for (int i = 0; i < G; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 f[2][4];
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned addr = base + (unsigned)(((lane & 31) * 128) + ((((lane >> 5)) ^ ((lane >> 1) & 7)) << 4));
  for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(f[0][i]) : "v"(addr + i * 4096));
  for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(f[1][i]) : "v"(addr + i * 4096));
  for (int b = 0; b < 2; ++b)     // the results are operands of the wait: their registers stay theirs until the loads landed
    for (int i = 0; i < 4; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[b][i])::"memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < R; ++r)
        asm volatile("ds_read_b128 %0, %1" : "=v"(f[(ks + 1) & 1][r & 3]) : "v"(addr + ((it + r) & 7) * 4096));
#pragma unroll
      for (int m = 0; m < G; ++m)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(f[ks][m & 1]), "v"(f[ks][2 + (m >> 1 & 1)]));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int b = 0; b < 2; ++b)
    for (int i = 0; i < 4; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[b][i])::"memory");
  float s = 0;
  for (int i = 0; i < G; ++i) s += acc[i][0];
  for (int b = 0; b < 2; ++b) for (int i = 0; i < 4; ++i) s += (float)f[b][i][0];
  if (s == 12345.f) sink[0] = s;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}
template <int NWAVES, int G, int R, int RANDOM = 0>
void runk(float* sink, unsigned long long* cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  hipFuncSetAttribute((const void*)kstep<NWAVES, G, R, RANDOM>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kstep<NWAVES, G, R, RANDOM>), dim3(256), dim3(NWAVES * 64), 65536, 0, iters, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = iters * 2.0 * G;
  printf("kstep%s: waves/CU %2d  %d MFMA + %d ds_read_b128 per k-step : %5.1f ticks/MFMA/wave  %5.2f ns/MFMA/SIMD  (tick rate %.2f GHz, %.0f TF/s chip)\n",
         RANDOM ? " (random operands)" : "", NWAVES, G, R, (double)c / n, ms * 1e6 / n / (NWAVES / 4.0), c / (ms * 1e6), 256.0 * NWAVES * n * 32768.0 / (ms * 1e-3) / 1e12);
}

// the same 64 x 64 wave tile on v_mfma_f32_16x16x32_bf16: 16 MFMAs on 16 accumulators + 8 ds_read_b128 per 32-deep k-step
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NWAVES, int R, int RANDOM>
__global__ __launch_bounds__(NWAVES * 64) void kstep16(int iters, float* sink, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += NWAVES * 64) {
    if (RANDOM) {
      unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      const unsigned lo = (h & 0x807fu) | 0x3b80u, hi = ((h >> 16) & 0x807fu) | 0x3b80u;
      reinterpret_cast<unsigned*>(smem)[i] = lo | (hi << 16);
    } else {
      reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 127);
    }
  }
  __syncthreads();
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  bf16x8 f[2][8];
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  // row lane%16 (128-B rows), 16-B chunk lane/16 of the k-step, XOR (row >> 1) & 7: conflict-free for the b128 lane groups
  unsigned addr = base + (unsigned)(((lane & 15) * 128) + ((((lane >> 4)) ^ ((lane >> 1) & 7)) << 4));
  for (int b = 0; b < 2; ++b)
    for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(f[b][i]) : "v"(addr + i * 2048));
  for (int b = 0; b < 2; ++b)
    for (int i = 0; i < 8; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[b][i])::"memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < R; ++r)
        asm volatile("ds_read_b128 %0, %1" : "=v"(f[(ks + 1) & 1][r & 7]) : "v"(addr + ((it + r) & 15) * 2048));
#pragma unroll
      for (int m = 0; m < 16; ++m)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(f[ks][m & 3]), "v"(f[ks][4 + (m >> 2)]));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int b = 0; b < 2; ++b)
    for (int i = 0; i < 8; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[b][i])::"memory");
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  for (int b = 0; b < 2; ++b) for (int i = 0; i < 8; ++i) s += (float)f[b][i][0];
  if (s == 12345.f) sink[0] = s;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}
template <int NWAVES, int R, int RANDOM>
void runk16(float* sink, unsigned long long* cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  hipFuncSetAttribute((const void*)kstep16<NWAVES, R, RANDOM>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kstep16<NWAVES, R, RANDOM>), dim3(256), dim3(NWAVES * 64), 65536, 0, iters, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = iters * 2.0 * 16;
  printf("kstep16x16x32%s: waves/CU %2d  16 MFMA + %d ds_read_b128 per k-step : %5.1f ticks/MFMA/wave  (tick rate %.2f GHz, %.0f TF/s chip)\n",
         RANDOM ? " (random operands)" : "", NWAVES, R, (double)c / n, c / (ms * 1e6), 256.0 * NWAVES * n * 16384.0 / (ms * 1e-3) / 1e12);
}

// Weights straight from L2 into registers: per k-step TM ds_read_b128 (activation fragments) + TN global 1-KB fragment
// loads (pre-packed weight fragments, D k-steps ahead) + TM*TN MFMAs.  Waves with the same wave % 4 read the same weights.
template <int NWAVES, int TM, int TN, int D>
__global__ __launch_bounds__(NWAVES * 64) void kstep_regb(int iters, const char* w, float* sink, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave & 3;
  for (int i = tid; i < 16384; i += NWAVES * 64) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned lo = (h & 0x807fu) | 0x3b80u, hi = ((h >> 16) & 0x807fu) | 0x3b80u;
    reinterpret_cast<unsigned*>(smem)[i] = lo | (hi << 16);
  }
  __syncthreads();
  constexpr int G = TM * TN, RB = D + 1, NBLK = 4 * TN;      // 1-KB blocks per k-step over the 4 column waves
  f32x16 acc[G];
  for (int i = 0; i < G; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 fa[2][TM], fb[RB][TN];
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned addr = base + (unsigned)(((lane & 31) * 128) + ((((lane >> 5)) ^ ((lane >> 1) & 7)) << 4));
  // An asm load whose result is never used leaves its registers free for the compiler while the load is still in
  // flight (it then lands on whatever was put there, an address for instance): every load below is either consumed
  // behind a wait or named as an operand of the wait that retires it.
  for (int i = 0; i < TM; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[0][i]) : "v"(addr + i * 4096));
  for (int i = 0; i < TM; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][i])::"memory");
  const unsigned voff = (unsigned)(wn * TN * 1024 + lane * 16);
  const unsigned KSTEPS = 144;                                // 9 taps x 256 channels / 16
  unsigned q = 0;                                             // k-step whose weights are requested next
  for (int d = 0; d < D; ++d) {
    const char* p = w + (size_t)q * NBLK * 1024;
    for (int j = 0; j < TN; ++j) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(fb[d][j]) : "v"(voff), "s"(p), "n"(0));
    q = q + 1 == KSTEPS ? 0 : q + 1;
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2 * RB; ++u) {
      const char* p = w + (size_t)q * NBLK * 1024;
      q = q + 1 == KSTEPS ? 0 : q + 1;
#pragma unroll
      for (int j = 0; j < TN; ++j)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(fb[(u + D) % RB][j]) : "v"(voff + j * 1024), "s"(p));
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(D * TN) : "memory");
#pragma unroll
      for (int r = 0; r < TM; ++r)
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa[(u + 1) & 1][r]) : "v"(addr + ((it + r) & 7) * 4096));
#pragma unroll
      for (int m = 0; m < G; ++m)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(fb[u % RB][m % TN]), "v"(fa[u & 1][m / TN]));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int d = 0; d < RB; ++d)
    for (int j = 0; j < TN; ++j) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(fb[d][j])::"memory");
  for (int b = 0; b < 2; ++b)
    for (int i = 0; i < TM; ++i) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(fa[b][i])::"memory");
  float s = 0;
  for (int i = 0; i < G; ++i) s += acc[i][0];
  for (int d = 0; d < RB; ++d) for (int j = 0; j < TN; ++j) s += (float)fb[d][j][0];
  for (int b = 0; b < 2; ++b) for (int i = 0; i < TM; ++i) s += (float)fa[b][i][0];
  if (s == 12345.f) sink[0] = s;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}
template <int NWAVES, int TM, int TN, int D>
void runb(const char* w, float* sink, unsigned long long* cyc) {
  const int iters = 500;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  hipFuncSetAttribute((const void*)kstep_regb<NWAVES, TM, TN, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kstep_regb<NWAVES, TM, TN, D>), dim3(256), dim3(NWAVES * 64), 65536, 0, iters, w, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = iters * 2.0 * (D + 1) * TM * TN;
  printf("regB: waves/CU %2d  %d MFMA + %d ds_read_b128 + %d global 1-KB loads (%d k-steps ahead) per k-step : %5.1f ticks/MFMA/wave  (%.0f TF/s chip)\n",
         NWAVES, TM * TN, TM, TN, D, (double)c / n, 256.0 * NWAVES * n * 32768.0 / (ms * 1e-3) / 1e12);
}

// The fused 256 x 256 tile's traffic in one loop, 8 waves, 12 k-steps (three stages) per period:
// MODE 0 -- weights through LDS: 8 MFMA + 6 ds_read_b128 per k-step; waves 0-3 send 8 weight pieces (1-KB LDS-DMA) at the top
//           of each stage and 9 window pieces at the top of the first; vmcnt wait + barrier at the end of every stage.
// MODE 1 -- weights from L2 into registers: 8 MFMA + 4 ds_read_b128 + 2 global 1-KB loads (3 k-steps ahead) per k-step;
//           waves 0-3 send one window piece in each of the first nine k-steps; one barrier per period.
template <int MODE, int STRIDED = 0>
__global__ __launch_bounds__(512) void kmix(int iters, const char* w, unsigned wbytes, float* sink, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave & 3;
  const bool stager = wave < 4;
  for (int i = tid; i < 8192; i += 512) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    reinterpret_cast<unsigned*>(smem)[i] = (h & 0x807fu) | 0x3b80u | ((((h >> 16) & 0x807fu) | 0x3b80u) << 16);
  }
  __syncthreads();
  constexpr int TM = 4, TN = 2, G = 8, D = 3, RB = D + 1, U = 12;
  f32x16 acc[G];
  for (int i = 0; i < G; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 fa[2][TM], fb[RB][TN];
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned addr = base + (unsigned)(((lane & 31) * 128) + ((((lane >> 5)) ^ ((lane >> 1) & 7)) << 4));
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)wbytes, 0x00020000);
  // DMA source cursor (wraps inside the 2 MB buffer).  STRIDED: a piece is 8 rows of 128 B at a 512-B row stride (a 64-channel
  // chunk of 256-channel rows, what the real tile fetches) instead of 1 KB in a row
  unsigned dsrc = STRIDED ? (unsigned)(wave * 9 * 4096 + (lane >> 3) * 512 + (lane & 7) * 16) : (unsigned)(wave * 9 * 1024 + lane * 16);
  unsigned ddst = 0;                                         // DMA destination piece inside [32 KB, 96 KB)
#define DMA_PIECE()                                                                                    \
  do {                                                                                                 \
    auto lds_ = (__attribute__((address_space(3))) void*)(smem + 32768 + ((ddst + wave) & 63) * 1024);   \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds_, 16, dsrc, 0, 0, 0);                             \
    dsrc = (dsrc + (STRIDED ? 36 * 4096 + 128 : 36 * 1024)) & (1u << 20) - 1;                          \
    ddst += 4;                                                                                         \
  } while (0)
  for (int i = 0; i < TM; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[0][i]) : "v"(addr + i * 4096));
  for (int i = 0; i < TM; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][i])::"memory");
  if (MODE == 0) {
    for (int j = 0; j < TN; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(fb[0][j]) : "v"(addr + 16384 + j * 4096));
    for (int j = 0; j < TN; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0][j])::"memory");
  }
  const unsigned voff = (unsigned)(wn * TN * 1024 + lane * 16);
  unsigned q = 0;
  if (MODE == 1) {
    for (int d = 0; d < D; ++d) {
      const char* p = w + (size_t)q * 8 * 1024;
      for (int j = 0; j < TN; ++j) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(fb[d][j]) : "v"(voff + j * 1024), "s"(p));
      q = q + 1 == 144 ? 0 : q + 1;
    }
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 0) {
        if (u % 4 == 0 && stager) {
#pragma unroll
          for (int k = 0; k < 8; ++k) DMA_PIECE();
          if (u == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) DMA_PIECE();
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < G; ++m) {
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(fb[u & 1][m % TN]), "v"(fa[u & 1][m / TN]));
          if (m < 3) {
#pragma unroll
            for (int r = 2 * m; r < 2 * m + 2; ++r) {
              if (r < TM) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[(u + 1) & 1][r]) : "v"(addr + ((it + r) & 3) * 4096));
              else asm volatile("ds_read_b128 %0, %1" : "=v"(fb[(u + 1) & 1][r - TM]) : "v"(addr + 16384 + ((it + r) & 3) * 4096));
            }
          }
        }
        if (u % 4 == 3) {
          if (u == 3) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
      } else {
        if (u < 9 && stager) DMA_PIECE();
        const char* p = w + (size_t)q * 8 * 1024;
        q = q + 1 == 144 ? 0 : q + 1;
#pragma unroll
        for (int j = 0; j < TN; ++j)
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(fb[(u + D) % RB][j]) : "v"(voff + j * 1024), "s"(p));
        // loads issued behind the weights of this k-step: those of the last D k-steps, and the stagers' pieces among them
        int np = 0;
#pragma unroll
        for (int v = u - D + 1; v <= u; ++v) np += ((v + U) % U) < 9 ? 1 : 0;
        if (stager) {
          if (np == 3) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
          else if (np == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
          else if (np == 1) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        }
        if (u == U - 1) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < G; ++m) {
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(fb[u % RB][m % TN]), "v"(fa[u & 1][m / TN]));
          if (m < 2) {
#pragma unroll
            for (int r = 2 * m; r < 2 * m + 2; ++r)
              asm volatile("ds_read_b128 %0, %1" : "=v"(fa[(u + 1) & 1][r]) : "v"(addr + ((it + r) & 7) * 4096));
          }
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int d = 0; d < RB; ++d)
    for (int j = 0; j < TN; ++j) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(fb[d][j])::"memory");
  for (int b = 0; b < 2; ++b)
    for (int i = 0; i < TM; ++i) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(fa[b][i])::"memory");
  float s = 0;
  for (int i = 0; i < G; ++i) s += acc[i][0];
  for (int d = 0; d < RB; ++d) for (int j = 0; j < TN; ++j) s += (float)fb[d][j][0];
  for (int b = 0; b < 2; ++b) for (int i = 0; i < TM; ++i) s += (float)fa[b][i][0];
  __syncthreads();
  if (s == 12345.f) sink[0] = s + smem[40000 + tid];
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}
template <int MODE, int STRIDED = 0>
void runmix(const char* w, float* sink, unsigned long long* cyc, int blocks) {
  const int iters = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  hipFuncSetAttribute((const void*)kmix<MODE, STRIDED>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kmix<MODE, STRIDED>), dim3(blocks), dim3(512), 98304, 0, iters, w, 2u << 20, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = iters * 12.0 * 8;
  printf("mix %s%s, %d workgroups: %.0f TF/s per 256 CUs (%.1f ticks/MFMA/wave)\n", STRIDED ? "[pieces of 8 x 128 B rows] " : "",
         MODE ? "weights L2 -> registers, window pieces by DMA" : "weights + window by DMA through LDS", blocks,
         256.0 * 8 * n * 32768.0 / (ms * 1e-3) / 1e12, (double)c / n);
}

// The same traffic on FOUR waves of 128 x 128 (one per SIMD, accumulators in AGPRs): 16 MFMA + 8 ds_read_b128 per k-step and
// wave, every wave sends 8 weight pieces per stage and 9 window pieces per three stages.  Half the LDS reads per MFMA of the
// 8-wave tile, but nothing else runs on a SIMD while its one wave issues DMA or waits.
__global__ __launch_bounds__(256) void kmix4(int iters, const char* w, unsigned wbytes, float* sink, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 8192; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    reinterpret_cast<unsigned*>(smem)[i] = (h & 0x807fu) | 0x3b80u | ((((h >> 16) & 0x807fu) | 0x3b80u) << 16);
  }
  __syncthreads();
  constexpr int TM = 4, TN = 4, G = 16, U = 12;
  f32x16 acc[G];
  for (int i = 0; i < G; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int i = 0; i < G; ++i) asm volatile("" : "+a"(acc[i]));
  bf16x8 fa[2][TM], fb[2][TN];
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned addr = base + (unsigned)(((lane & 31) * 128) + ((((lane >> 5)) ^ ((lane >> 1) & 7)) << 4));
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)wbytes, 0x00020000);
  unsigned dsrc = (unsigned)(wave * 9 * 1024 + lane * 16);
  unsigned ddst = 0;
#define DMA_PIECE4()                                                                                   \
  do {                                                                                                 \
    auto lds_ = (__attribute__((address_space(3))) void*)(smem + 32768 + ((ddst + wave) & 63) * 1024);   \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds_, 16, dsrc, 0, 0, 0);                             \
    dsrc = (dsrc + 36 * 1024) & (1u << 20) - 1;                                                        \
    ddst += 4;                                                                                         \
  } while (0)
  for (int i = 0; i < TM; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[0][i]) : "v"(addr + i * 4096));
  for (int j = 0; j < TN; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(fb[0][j]) : "v"(addr + 16384 + j * 4096));
  for (int i = 0; i < TM; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][i])::"memory");
  for (int j = 0; j < TN; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0][j])::"memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int m = 0; m < G; ++m) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(fb[u & 1][m % TN]), "v"(fa[u & 1][m / TN]));
        if (m < 4) {
#pragma unroll
          for (int r = 2 * m; r < 2 * m + 2; ++r) {
            if (r < TM) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[(u + 1) & 1][r]) : "v"(addr + ((it + r) & 3) * 4096));
            else asm volatile("ds_read_b128 %0, %1" : "=v"(fb[(u + 1) & 1][r - TM]) : "v"(addr + 16384 + ((it + r) & 3) * 4096));
          }
        }
        // the stage's DMA pieces go out one per MFMA behind the fragment reads (the wave is the SIMD's only one: a burst in
        // front of the MFMAs would idle the pipe for its whole issue time)
        if (u % 4 == 0 && m >= 4 && m < 12) DMA_PIECE4();
        if (u == 1 && m >= 4 && m < 13) DMA_PIECE4();
      }
      if (u % 4 == 3) {
        if (u == 3) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int b = 0; b < 2; ++b) {
    for (int i = 0; i < TM; ++i) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(fa[b][i])::"memory");
    for (int j = 0; j < TN; ++j) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(fb[b][j])::"memory");
  }
  float s = 0;
  for (int i = 0; i < G; ++i) s += acc[i][0];
  for (int b = 0; b < 2; ++b) for (int i = 0; i < TM; ++i) s += (float)fa[b][i][0] + (float)fb[b][i][0];
  __syncthreads();
  if (s == 12345.f) sink[0] = s + smem[40000 + tid];
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}
void runmix4(const char* w, float* sink, unsigned long long* cyc, int blocks) {
  const int iters = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  hipFuncSetAttribute((const void*)kmix4, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kmix4, dim3(blocks), dim3(256), 98304, 0, iters, w, 2u << 20, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = iters * 12.0 * 16;
  printf("mix 4 waves of 128 x 128 (AGPR accumulators), weights + window by DMA through LDS, %d workgroups: %.0f TF/s per 256 CUs (%.1f ticks/MFMA/wave)\n",
         blocks, 256.0 * 4 * n * 32768.0 / (ms * 1e-3) / 1e12, (double)c / n);
}

// fp32 matrix rate: v_mfma_f32_32x32x2_f32 back to back on 4 accumulators, W waves per SIMD, operands in registers
template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void kf32(int iters, float* sink, unsigned long long* cyc) {
  const int tid = threadIdx.x;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  unsigned h = (unsigned)tid * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = __uint_as_float(((h >> i) & 0x807fffffu) | 0x3b800000u); b[i] = __uint_as_float(((h << i) & 0x807fffffu) | 0x3b000000u); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m)
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a[m >> 2]), "v"(b[m & 3]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  if (s == 12345.f) sink[0] = s;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}
template <int NWAVES>
void runf32(float* sink, unsigned long long* cyc) {
  const int iters = 4000 / (NWAVES / 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kf32<NWAVES>), dim3(256), dim3(NWAVES * 64), 0, 0, iters, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double n = (double)iters * 16;
  printf("fp32 32x32x2 MFMA, %2d waves/CU: %.1f TF/s chip (%.2f ms)\n", NWAVES, 256.0 * NWAVES * n * 4096.0 / (ms * 1e-3) / 1e12, ms);
}

int main(int argc, char**) {
  float* sink; unsigned long long* cyc;
  hipMalloc(&sink, 4); hipMalloc(&cyc, 8);
  char* wbuf; hipMalloc(&wbuf, 144 * 16 * 1024 + 65536); hipMemset(wbuf, 0x3b, 144 * 16 * 1024 + 65536);
  for (int i = 0; i < 40; ++i)        // the clock takes a while to come up on a cold card
    hipLaunchKernelGGL((kstep<8, 8, 0, 1>), dim3(256), dim3(512), 65536, 0, 2000, sink, cyc);
  hipDeviceSynchronize();
  runf32<4>(sink, cyc); runf32<8>(sink, cyc); runf32<16>(sink, cyc); runf32<4>(sink, cyc);
  for (int rep = 0; rep < 3; ++rep) {
    runmix<0, 1>(wbuf, sink, cyc, 256); runmix<0, 1>(wbuf, sink, cyc, 127);
    runmix<0>(wbuf, sink, cyc, 256); runmix<1>(wbuf, sink, cyc, 256); runmix<0>(wbuf, sink, cyc, 127); runmix<1>(wbuf, sink, cyc, 127);
    runk<8, 8, 6, 1>(sink, cyc);
    runb<8, 4, 2, 2>(wbuf, sink, cyc); runb<8, 4, 2, 3>(wbuf, sink, cyc); runb<8, 4, 2, 5>(wbuf, sink, cyc);
    runb<8, 2, 4, 3>(wbuf, sink, cyc); runb<4, 4, 2, 3>(wbuf, sink, cyc); runb<8, 2, 2, 3>(wbuf, sink, cyc);
  }
  if (argc < 2) return 0;      // any argument: the older sweeps too
  hipFuncSetAttribute((const void*)probe<4, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  run<4, 0, 0>(sink, cyc); run<4, 2, 0>(sink, cyc); run<4, 4, 0>(sink, cyc); run<4, 6, 0>(sink, cyc); run<4, 8, 0>(sink, cyc);
  run<4, 0, 8>(sink, cyc); run<4, 0, 16>(sink, cyc); run<4, 0, 32>(sink, cyc);
  run<4, 6, 16>(sink, cyc); run<4, 6, 24>(sink, cyc);
  run<8, 0, 0>(sink, cyc); run<8, 4, 0>(sink, cyc); run<8, 6, 16>(sink, cyc); run<8, 8, 16>(sink, cyc);
  run<1, 6, 16>(sink, cyc);
  runk<4, 4, 0>(sink, cyc); runk<4, 4, 4>(sink, cyc); runk<8, 4, 0>(sink, cyc); runk<8, 4, 4>(sink, cyc);
  runk<8, 4, 2>(sink, cyc); runk<4, 8, 6>(sink, cyc); runk<8, 8, 6>(sink, cyc); runk<4, 8, 4>(sink, cyc);
  runk<8, 8, 0>(sink, cyc);
  runk<4, 4, 0, 1>(sink, cyc); runk<8, 4, 0, 1>(sink, cyc); runk<8, 4, 4, 1>(sink, cyc); runk<4, 8, 6, 1>(sink, cyc); runk<8, 8, 6, 1>(sink, cyc);
  // same FLOPs and LDS bytes per k-step pair: 32x32x16 (4 MFMA + 4 reads, twice) vs 16x16x32 (16 MFMA + 8 reads)
  runk<8, 4, 4, 1>(sink, cyc); runk16<8, 8, 1>(sink, cyc); runk<8, 4, 4, 1>(sink, cyc); runk16<8, 8, 1>(sink, cyc);
  runk<4, 4, 4, 1>(sink, cyc); runk16<4, 8, 1>(sink, cyc); runk16<8, 8, 0>(sink, cyc); runk16<8, 0, 1>(sink, cyc);
  return 0;
}
