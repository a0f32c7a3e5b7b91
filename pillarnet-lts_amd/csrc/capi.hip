// C-ABI plumbing shared by every entry point: version + thread-local error text.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void pn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int pn_abi_version(void) { return 1; }
extern "C" const char* pn_last_error(void) { return g_err; }

// Device memory fill on a stream, as a kernel.  The entry points clear their scratch with this instead of
// hipMemsetAsync so that a caller may capture them into a HIP graph: memset nodes have replayed wrongly on
// this platform (wrong results on replays after the first).  byte is replicated; 16-byte stores over the
// aligned middle, byte stores at the ragged ends.
__global__ __launch_bounds__(256) void pn_fill_kernel(unsigned char* p, unsigned char byte, size_t head,
                                                      size_t vecs, size_t tail_start, size_t bytes) {
  const unsigned w = 0x01010101u * byte;
  const uint4 v = make_uint4(w, w, w, w);
  uint4* mid = reinterpret_cast<uint4*>(p + head);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < vecs; i += (size_t)gridDim.x * 256) mid[i] = v;
  if (blockIdx.x == 0) {
    for (size_t i = threadIdx.x; i < head; i += 256) p[i] = byte;
    for (size_t i = tail_start + threadIdx.x; i < bytes; i += 256) p[i] = byte;
  }
}

int pn_fill_async(void* ptr, int byte, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return PN_OK;
  unsigned char* p = (unsigned char*)ptr;
  size_t head = (16 - ((uintptr_t)p & 15)) & 15;
  if (head > bytes) head = bytes;
  const size_t vecs = (bytes - head) / 16, tail_start = head + vecs * 16;
  size_t blocks = (vecs + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(pn_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, (unsigned char)byte, head, vecs,
                     tail_start, bytes);
  PN_CHECK_LAUNCH("pn_fill_kernel");
  return PN_OK;
}
