// Micro-probe: cost of LDS-DMA staging per stage on one workgroup per CU.
//   stage = every wave issues P pieces (buffer_load_dwordx4 ... lds, 1 KiB each) from an L2-resident
//   region shared by all CUs, then waits (depth 1: vmcnt(0); depth 2: the previous stage's pieces) and barriers.
// Build: hipcc -O3 --offload-arch=gfx950 dma_probe.hip -o dma_probe ; run: ./dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// MODE 0: contiguous 1 KiB pieces.  1: 8 rows x 128 B at a 512-B row stride.  2: + XOR chunk swizzle.
// 3: + per-row index read from LDS (gather).  4: mode 2 with a 2048-B row stride (cin = 1024-like).
template <int NW, int P, int DEPTH, int MODE>
__global__ __launch_bounds__(NW * 64) void probe(const char* src, unsigned bytes, int stages, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
  constexpr int STAGE_BYTES = NW * P * 1024;
  int* idx = reinterpret_cast<int*>(smem + 3 * STAGE_BYTES);
  if (MODE == 3) {
    for (int q = tid; q < NW * P * 8; q += NW * 64) idx[q] = (q * 37) % 4000;   // scattered rows of a 2 MiB / 512 B table
    __syncthreads();
  }
  for (int s = 0; s < stages; ++s) {
    char* base = smem + (s % 3) * STAGE_BYTES;
    const unsigned soff = (unsigned)((s * STAGE_BYTES) % (bytes - STAGE_BYTES));
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int piece = wave * P + j;
      auto lds = (__attribute__((address_space(3))) void*)(base + piece * 1024);
      unsigned off;
      if (MODE == 0) off = soff + piece * 1024 + lane * 16;
      else {
        const int row = piece * 8 + lane / 8, slot = lane % 8;
        const unsigned stride = MODE == 4 ? 2048u : 512u;
        const int key = MODE >= 2 ? (row >> 1) & 7 : 0;
        const unsigned r = MODE == 3 ? (unsigned)idx[row] : (unsigned)row;
        off = (r * stride + (unsigned)((s & 3) * 128) + (unsigned)((slot ^ key) << 4)) % (bytes - 1024u);
        off &= ~15u;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds, 16, off, 0, 0, 0);
    }
    if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && smem[lane] == 77) sink[0] = 1;
}

template <int NW, int P, int DEPTH, int MODE = 0>
void run(const char* src, unsigned bytes, int* sink) {
  const int stages = 400;
  const size_t lds = 3 * (size_t)NW * P * 1024 + (size_t)NW * P * 8 * 4;
  if (lds > 160 * 1024) return;
  hipFuncSetAttribute((const void*)probe<NW, P, DEPTH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NW, P, DEPTH, MODE>), dim3(256), dim3(NW * 64), lds, 0, src, bytes, stages, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us_stage = ms * 1e3 / stages;
  printf("mode %d waves %d pieces/wave %2d depth %d: %6.3f us/stage  %5.1f KiB/stage/CU  %6.1f GB/s/CU  (%5.0f cyc/stage @2.4GHz)\n",
         MODE, NW, P, DEPTH, us_stage, NW * P * 1.0, NW * P * 1024 / us_stage * 1e-3, us_stage * 2400);
}

int main() {
  const unsigned bytes = 2u << 20;   // 2 MiB: L2-resident, read by every CU
  char* src; int* sink;
  hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&sink, 4);
  run<8, 6, 1, 0>(src, bytes, sink); run<8, 6, 1, 1>(src, bytes, sink); run<8, 6, 1, 2>(src, bytes, sink);
  run<8, 6, 1, 3>(src, bytes, sink); run<8, 6, 1, 4>(src, bytes, sink);
  run<8, 6, 2, 0>(src, bytes, sink); run<8, 6, 2, 1>(src, bytes, sink); run<8, 6, 2, 2>(src, bytes, sink);
  run<8, 6, 2, 3>(src, bytes, sink); run<8, 6, 2, 4>(src, bytes, sink);
  run<4, 12, 1, 2>(src, bytes, sink); run<4, 12, 1, 3>(src, bytes, sink); run<4, 12, 2, 3>(src, bytes, sink);
  run<8, 4, 1, 2>(src, bytes, sink); run<8, 4, 1, 3>(src, bytes, sink);
  return 0;
}
