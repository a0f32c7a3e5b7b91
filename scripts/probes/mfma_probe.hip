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
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
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
  float s = 0;
  for (int i = 0; i < G; ++i) s += acc[i][0];
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
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
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

int main() {
  float* sink; unsigned long long* cyc;
  hipMalloc(&sink, 4); hipMalloc(&cyc, 8);
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
