// (a9/a10) CenterHead.predict + post_processing + rotate_nms_pcdet's pre-NMS part, on the device.
//
// Reference being replaced (~40 tiny torch kernels, boolean-index syncs and a full sort per task
// and sample):
//   det3d/models/bbox_heads/center_head.py:217-350   predict: sigmoid / exp(clamp) / atan2 / grid
//   det3d/models/bbox_heads/center_head.py:353-413   post_processing: class max, score & range mask
//   det3d/core/bbox/box_torch_ops.py:296-322         rotate_nms_pcdet: rectified score, sort desc,
//                                                    top pre_max, to_pcdet, [:post_max]
//   det3d/core/bbox/box_torch_ops.py:325-359         rotate_class_nms_pcdet (per-class sets)
//   det3d/ops/iou3d_nms/iou3d_nms_utils.py:30-34     to_pcdet
//
// A "set" is one NMS problem: (task, optional class filter) x sample.
//   decode_score_kernel   all cells of all sets in parallel: score, label, masks -> a 32-bit sort
//                         key per cell (0 = rejected) + a per-set histogram of the key range.
//   decode_select_kernel  one 1024-thread workgroup per set: histogram suffix scan -> cut-off bin,
//                         collect the cells at or above it into LDS, bitonic sort by
//                         (score desc, cell asc), keep the first pre_max, decode their boxes.
//   (nms.hip runs on the sorted boxes)
//   gather_detections_kernel  per sample: concatenates the kept boxes of its sets in set order.
#include "common.h"
#include <math.h>

static constexpr int DEC_BINS = 8192;
static constexpr int DEC_LIST = 8192;   // LDS candidate slots per set (sure < pre_max <= 4096, + cut-off bin)
static constexpr int SEL_THREADS = 1024;

struct DecodeSetDev {
  const float* maps;     // [B, H*W, CH]
  int H, W, CH, LD, ncls, has_vel, has_iou, class_filter, label_offset, stride;
  float score_thr, rectifier, nms_thr;
  int use_rectify, pre_max, post_max;
  long long key_offset;  // start of this set-group's keys in the key scratch (per sample: + b*H*W)
  int hm_only;           // maps hold the heat-map channels only (regression rows are evaluated later, at the
                         // selected cells): no range mask here, hm starts at channel 0
};

struct DecodeCommonDev {
  int B, num_groups;     // groups = tasks x class filters; sets = groups x B, set = group*B + b
  float pillar_size, x_min, y_min;
  float range[6];
  int has_range;
  unsigned key_lo;       // bits(min score threshold) -- histogram origin
  int key_shift;
};

#define PN_MAX_GROUPS 16
struct DecodeGroups {
  DecodeSetDev g[PN_MAX_GROUPS];
};

// channel layout of `maps`: reg(2) height(1) dim(3) rot(2) [vel(2)] [iou(1)] hm(ncls)
__device__ __forceinline__ int ch_vel(const DecodeSetDev& s) { return 8; }
__device__ __forceinline__ int ch_iou(const DecodeSetDev& s) { return 8 + 2 * s.has_vel; }
__device__ __forceinline__ int ch_hm(const DecodeSetDev& s) { return 8 + 2 * s.has_vel + s.has_iou; }

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float centre_coord(int j, float reg, int stride, float ps, float pmin) {
  // center_head.py:311-315: ((j + reg) * stride) * pillar_size + range_min, each op rounded
  float v = (float)j + reg;
  v = v * (float)stride;
  v = v * ps;
  return v + pmin;
}

__global__ __launch_bounds__(256) void decode_score_kernel(DecodeGroups G, DecodeCommonDev C,
                                                           unsigned* __restrict__ keys,
                                                           unsigned* __restrict__ hist) {
  // block-private histogram: scores of a random-init or saturated head pile into a few bins, and
  // 10^5 global atomics on one address serialise (measured 0.5 ms); LDS atomics do not
  __shared__ unsigned lhist[DEC_BINS];
  const int group = blockIdx.y, b = blockIdx.z;
  const DecodeSetDev& S = G.g[group];
  const int cells = S.H * S.W;
  const int set = group * C.B + b;
  for (int i = threadIdx.x; i < DEC_BINS; i += 256) lhist[i] = 0u;
  __syncthreads();
  for (int cell = blockIdx.x * 256 + threadIdx.x; cell < cells; cell += gridDim.x * 256) {
    const float* m = S.maps + ((size_t)b * cells + cell) * S.LD;
    const int hm0 = S.hm_only ? 0 : ch_hm(S);
    float best = sigmoidf_ref(m[hm0]);
    int label = 0;
    for (int c = 1; c < S.ncls; ++c) {
      const float v = sigmoidf_ref(m[hm0 + c]);
      if (v > best) { best = v; label = c; }
    }
    bool ok = best > S.score_thr;
    if (S.class_filter >= 0) ok = ok && label == S.class_filter;
    if (ok && C.has_range && !S.hm_only) {
      const int yy = cell / S.W, xx = cell - yy * S.W;
      const float x = centre_coord(xx, m[0], S.stride, C.pillar_size, C.x_min);
      const float y = centre_coord(yy, m[1], S.stride, C.pillar_size, C.y_min);
      const float z = m[2];
      ok = x >= C.range[0] && y >= C.range[1] && z >= C.range[2] && x <= C.range[3] &&
           y <= C.range[4] && z <= C.range[5];
    }
    unsigned key = 0;
    if (ok) {
      float rect = best;
      if (S.rectifier != 0.f) {
        float iou = 1.f;
        if (S.has_iou) {
          iou = (m[ch_iou(S)] + 1.f) * 0.5f;       // center_head.py:264-267
          iou = fminf(fmaxf(iou, 0.f), 1.f);
        }
        rect = powf(best, 1.f - S.rectifier) * powf(iou, S.rectifier);
      }
      key = __float_as_uint(rect);
      if ((int)key <= 0) key = 1;  // non-positive / NaN scores cannot outrank anything
      int bin = key > C.key_lo ? (int)((key - C.key_lo) >> C.key_shift) : 0;
      if (bin >= DEC_BINS) bin = DEC_BINS - 1;
      atomicAdd(&lhist[bin], 1u);
    }
    keys[S.key_offset + (long long)b * cells + cell] = key;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < DEC_BINS; i += 256) {
    const unsigned v = lhist[i];
    if (v) atomicAdd(&hist[(size_t)set * DEC_BINS + i], v);
  }
}

// Descending bitonic sort of E * SEL_THREADS entries held E per thread in registers (thread t owns entries
// t*E .. t*E+E-1).  A compare-exchange with a partner in the same thread is a register swap, in the same
// wavefront a shuffle; only partners in another wavefront (distance >= 64*E) go through LDS and a barrier --
// 10 of the 55-91 passes.  The plain LDS version below spent ~1000 cycles per pass on its barrier.
template <int E>
__device__ __forceinline__ void bitonic_sort_regs(unsigned long long* a) {
  const int t = threadIdx.x;
  unsigned long long v[E];
#pragma unroll
  for (int r = 0; r < E; ++r) v[r] = a[t * E + r];
  constexpr int P = E * SEL_THREADS;
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64 * E) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; ++r) a[t * E + r] = v[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const int e = t * E + r;
          const unsigned long long y = a[e ^ j];
          const bool keep_max = ((e & k) == 0) == ((e & j) == 0);
          v[r] = keep_max ? (v[r] > y ? v[r] : y) : (v[r] < y ? v[r] : y);
        }
      } else if (j >= E) {
        const int lane_xor = j / E;
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const int e = t * E + r;
          const unsigned lo = __shfl_xor((unsigned)(v[r] & 0xFFFFFFFFull), lane_xor);
          const unsigned hi = __shfl_xor((unsigned)(v[r] >> 32), lane_xor);
          const unsigned long long y = ((unsigned long long)hi << 32) | lo;
          const bool keep_max = ((e & k) == 0) == ((e & j) == 0);
          v[r] = keep_max ? (v[r] > y ? v[r] : y) : (v[r] < y ? v[r] : y);
        }
      } else {
        // same thread: j is 1, 2 or 4 here; every register index below is a compile-time constant (a run-time
        // index would move v[] to scratch memory)
#pragma unroll
        for (int J = 1; J < E; J <<= 1) {
          if (j == J) {
#pragma unroll
            for (int r = 0; r < E; ++r) {
              if ((r & J) == 0) {
                const bool desc = (((t * E + r) & k) == 0);
                const unsigned long long x = v[r], y = v[r | J];
                const unsigned long long mx = x > y ? x : y, mn = x > y ? y : x;
                v[r] = desc ? mx : mn;
                v[r | J] = desc ? mn : mx;
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < E; ++r) a[t * E + r] = v[r];
  __syncthreads();
}

__device__ __forceinline__ void bitonic_sort_desc(unsigned long long* a, int n /*pow2*/) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += SEL_THREADS) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long x = a[i], y = a[p];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) { a[i] = y; a[p] = x; }
        }
      }
      __syncthreads();
    }
  }
}

// -DPN_DECODE_STAMPS (tuning builds only): s_memtime at the phase boundaries of set 0's workgroup
#ifdef PN_DECODE_STAMPS
__device__ unsigned long long dec_stamps[16];
extern "C" int pn_decode_debug_stamps(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dec_stamps), (size_t)(n > 16 ? 16 : n) * 8);
}
#define DEC_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) dec_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DEC_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(SEL_THREADS) void decode_select_kernel(
    DecodeGroups G, DecodeCommonDev C, const unsigned* __restrict__ keys,
    const unsigned* __restrict__ hist, int pre_cap, int box_dim, float* __restrict__ nms_boxes,
    float* __restrict__ cand_boxes, float* __restrict__ cand_scores,
    int32_t* __restrict__ cand_labels, int32_t* __restrict__ counts, float* __restrict__ nms_thr,
    int32_t* __restrict__ nms_post, int32_t* __restrict__ status, int32_t* __restrict__ sel_cells,
    int32_t* __restrict__ totals) {
  __shared__ unsigned long long list[DEC_LIST];
  __shared__ int sfx[SEL_THREADS];
  __shared__ int s_cut, s_cnt, s_above, s_incut, s_sub;
  const int set = blockIdx.x, group = set / C.B, b = set - group * C.B;
  const DecodeSetDev& S = G.g[group];
  const int cells = S.H * S.W, tid = threadIdx.x;
  int pre_max = S.pre_max < pre_cap ? S.pre_max : pre_cap;
  DEC_STAMP(0);

  // ---- cut-off bin: the highest bin t such that #keys in bins >= t reaches pre_max -----------
  constexpr int BPT = DEC_BINS / SEL_THREADS;  // bins per thread, thread 0 owns the TOP bins
  const unsigned* h = hist + (size_t)set * DEC_BINS;
  int mine = 0;
#pragma unroll
  for (int k = 0; k < BPT; ++k) mine += (int)h[DEC_BINS - 1 - (tid * BPT + k)];
  sfx[tid] = mine;
  __syncthreads();
  for (int d = 1; d < SEL_THREADS; d <<= 1) {  // inclusive scan from the top bin downwards
    const int v = tid >= d ? sfx[tid - d] : 0;
    __syncthreads();
    sfx[tid] += v;
    __syncthreads();
  }
  if (tid == 0) { s_cut = 0; s_cnt = 0; s_above = 0; s_incut = 0; s_sub = -1; }
  if (tid == SEL_THREADS - 1 && totals) totals[set] = sfx[tid];   // every candidate of the set
  // circle NMS ranks every cell above the threshold in the reference; here the best pre_max: say so when that cuts
  if (tid == SEL_THREADS - 1 && S.nms_thr < 0.f && sfx[tid] > pre_max) atomicOr(&status[0], 4);
  __syncthreads();
  {
    const int above = sfx[tid] - mine;  // keys in bins above this thread's bins
    if (above < pre_max && sfx[tid] >= pre_max) {
      int run = above;
      for (int k = 0; k < BPT; ++k) {
        const int bin = DEC_BINS - 1 - (tid * BPT + k);
        const int hb = (int)h[bin];
        run += hb;
        if (run >= pre_max) { s_cut = bin; s_above = run - hb; s_incut = hb; break; }
      }
    }
  }
  __syncthreads();
  const int cut = s_cut;  // 0 when the set holds fewer than pre_max candidates
  DEC_STAMP(1);
  // candidates are the keys >= klo; those >= khi lie above the cut-off range (fewer than pre_max of them)
  unsigned long long klo = cut == 0 ? 1ull : (unsigned long long)C.key_lo + ((unsigned long long)cut << C.key_shift);
  unsigned long long khi = cut == DEC_BINS - 1 ? (1ull << 32)
                                               : (unsigned long long)C.key_lo + ((unsigned long long)(cut + 1) << C.key_shift);
  const unsigned* kk = keys + S.key_offset + (long long)b * cells;

  // ---- a crowded cut-off bin (scores packed into a narrow band) is split once more, on the key bits below
  // the bin width, so that the candidate list stays near pre_max and the sort below stays short
  constexpr int SUB_BITS = 12, SUBS = 1 << SUB_BITS;
  if (cut > 0 && cut < DEC_BINS - 1 && C.key_shift > 0 && s_above + s_incut > pre_max + 1024) {
    int* sub = reinterpret_cast<int*>(list);
    const int sshift = C.key_shift > SUB_BITS ? C.key_shift - SUB_BITS : 0;
    const int nsub = 1 << (C.key_shift - sshift);
    for (int i = tid; i < SUBS; i += SEL_THREADS) sub[i] = 0;
    __syncthreads();
    for (int base = 0; base < cells; base += SEL_THREADS * 8) {
      unsigned kv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int cell = base + u * SEL_THREADS + tid;
        kv[u] = cell < cells ? kk[cell] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (kv[u] >= klo && kv[u] < khi) atomicAdd(&sub[(int)((kv[u] - (unsigned)klo) >> sshift)], 1);
    }
    __syncthreads();
    // suffix counts from the top sub-bin: thread t owns sub-bins nsub-1-4t .. nsub-4-4t
    constexpr int SPT = SUBS / SEL_THREADS;
    int smine = 0;
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
      const int sb = nsub - 1 - (tid * SPT + k);
      smine += sb >= 0 ? sub[sb] : 0;
    }
    sfx[tid] = smine;
    __syncthreads();
    for (int d = 1; d < SEL_THREADS; d <<= 1) {
      const int v = tid >= d ? sfx[tid - d] : 0;
      __syncthreads();
      sfx[tid] += v;
      __syncthreads();
    }
    const int need = pre_max - s_above;   // >= 1 candidates still wanted from the cut-off bin
    const int sabove = sfx[tid] - smine;
    if (sabove < need && sfx[tid] >= need) {
      int run = sabove;
      for (int k = 0; k < SPT; ++k) {
        const int sb = nsub - 1 - (tid * SPT + k);
        run += sb >= 0 ? sub[sb] : 0;
        if (run >= need) { s_sub = sb; break; }
      }
    }
    __syncthreads();
    if (s_sub >= 0) {
      klo = klo + ((unsigned long long)s_sub << sshift);
      khi = klo + (1ull << sshift);
    }
    __syncthreads();   // `list` is reused below
  }

  DEC_STAMP(2);
  // ---- collect ----------------------------------------------------------------------------------
  // keys are fetched 8 at a time per thread (independent loads in flight) -- a load-test-append loop
  // serialises one L2 round trip per cell and was most of this kernel's time
  for (int base = 0; base < cells; base += SEL_THREADS * 8) {
    unsigned kv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int cell = base + u * SEL_THREADS + tid;
      kv[u] = cell < cells ? kk[cell] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      // one LDS atomic per wavefront, not per hit: thousands of atomics on the one counter serialise
      const unsigned key = kv[u];
      const bool hit = key != 0u && key >= klo;
      const unsigned long long m = __ballot(hit);
      if (m == 0ull) continue;   // wave-uniform
      const int lane = tid & 63, leader = __builtin_ctzll(m);
      int wbase = 0;
      if (lane == leader) wbase = atomicAdd(&s_cnt, __popcll(m));
      wbase = __shfl(wbase, leader);
      if (hit) {
        const int cell = base + u * SEL_THREADS + tid;
        const int pos = wbase + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < DEC_LIST)
          list[pos] = ((unsigned long long)key << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)cell);
      }
    }
  }
  __syncthreads();
  int n = s_cnt;
  DEC_STAMP(3);
  if (n > DEC_LIST) {
    // More cells inside the cut-off range than the list holds (maps full of equal scores): report it,
    // and collect again by a fixed rule -- every key above the cut-off range, then the range's cells in
    // ascending cell order until the list is full -- so that the result does not depend on which thread's
    // atomicAdd came first.  Only this degenerate case pays for the ordered pass.
    if (tid == 0) { atomicOr(&status[0], 1); s_cnt = 0; }
    __syncthreads();
    for (int cell = tid; cell < cells; cell += SEL_THREADS) {
      const unsigned key = kk[cell];
      if (!key) continue;
      if (key >= khi) {   // fewer than pre_max of these by the definition of the cut-off range
        const int pos = atomicAdd(&s_cnt, 1);
        list[pos] = ((unsigned long long)key << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)cell);
      }
    }
    __syncthreads();
    int filled = s_cnt;              // same value in every thread
    // eight consecutive cells per thread and iteration: the block-wide prefix of the per-thread hit counts
    // (wave shuffles, then the 16 wave totals) ranks every hit in ascending cell order
    for (int base = 0; base < cells && filled < DEC_LIST; base += SEL_THREADS * 8) {
      const int c0 = base + tid * 8;
      unsigned kv[8];
      int mine = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int cell = c0 + u;
        unsigned key = cell < cells ? kk[cell] : 0u;
        if (!(key != 0u && key >= klo && key < khi)) key = 0u;
        kv[u] = key;
        mine += key != 0u;
      }
      const int lane = tid & 63, wv = tid >> 6;
      int incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
      }
      if (lane == 63) sfx[wv] = incl;
      __syncthreads();
      int before = 0, total = 0;
      for (int w = 0; w < SEL_THREADS / 64; ++w) {
        const int c = sfx[w];
        before += w < wv ? c : 0;
        total += c;
      }
      int pos = filled + before + incl - mine;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (kv[u]) {
          if (pos < DEC_LIST)
            list[pos] = ((unsigned long long)kv[u] << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)(c0 + u));
          ++pos;
        }
      }
      filled += total;
      __syncthreads();
    }
    n = filled < DEC_LIST ? filled : DEC_LIST;
  }
  DEC_STAMP(4);
  int p2 = SEL_THREADS;              // one entry per thread at least; the padding (0) sorts to the end
  while (p2 < n) p2 <<= 1;
  for (int i = n + tid; i < p2; i += SEL_THREADS) list[i] = 0ull;
  __syncthreads();
  static_assert(DEC_LIST == 8 * SEL_THREADS, "sort dispatch covers 1, 2, 4 and 8 entries per thread");
  if (p2 == SEL_THREADS) bitonic_sort_regs<1>(list);
  else if (p2 == 2 * SEL_THREADS) bitonic_sort_regs<2>(list);
  else if (p2 == 4 * SEL_THREADS) bitonic_sort_regs<4>(list);
  else bitonic_sort_regs<8>(list);
  DEC_STAMP(5);
  if (blockIdx.x == 0 && threadIdx.x == 0) { DEC_STAMP(7); }
#ifdef PN_DECODE_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) { dec_stamps[8] = (unsigned long long)n; dec_stamps[9] = (unsigned long long)p2; dec_stamps[10] = (unsigned long long)s_sub + 1; }
#endif
  const int m = n < pre_max ? n : pre_max;
  if (tid == 0) {
    counts[set] = m;
    nms_thr[set] = S.nms_thr;
    nms_post[set] = S.post_max < pre_cap ? S.post_max : pre_cap;
  }

  if (S.hm_only) {
    // ---- heat map only: hand the sorted cells on; their regression rows do not exist yet ----------
    for (int i = tid; i < m; i += SEL_THREADS) {
      const unsigned long long e = list[i];
      const int cell = (int)(0xFFFFFFFFu - (unsigned)(e & 0xFFFFFFFFull));
      const float* mp = S.maps + ((size_t)b * cells + cell) * S.LD;
      float best = sigmoidf_ref(mp[0]);
      int label = 0;
      for (int c = 1; c < S.ncls; ++c) {
        const float v = sigmoidf_ref(mp[c]);
        if (v > best) { best = v; label = c; }
      }
      sel_cells[(size_t)set * pre_cap + i] = cell;
      cand_scores[(size_t)set * pre_cap + i] = best;
      cand_labels[(size_t)set * pre_cap + i] = label + S.label_offset;
    }
    return;
  }
  // ---- decode the selected cells ---------------------------------------------------------------
  for (int i = tid; i < m; i += SEL_THREADS) {
    const unsigned long long e = list[i];
    const int cell = (int)(0xFFFFFFFFu - (unsigned)(e & 0xFFFFFFFFull));
    const float* mp = S.maps + ((size_t)b * cells + cell) * S.LD;
    const int yy = cell / S.W, xx = cell - yy * S.W;
    const float x = centre_coord(xx, mp[0], S.stride, C.pillar_size, C.x_min);
    const float y = centre_coord(yy, mp[1], S.stride, C.pillar_size, C.y_min);
    const float z = mp[2];
    const float d0 = expf(fminf(fmaxf(mp[3], -1.2f), 3.2f));
    const float d1 = expf(fminf(fmaxf(mp[4], -1.2f), 3.2f));
    const float d2 = expf(fminf(fmaxf(mp[5], -1.2f), 3.2f));
    const float rot = atan2f(mp[6], mp[7]);
    const int hm0 = ch_hm(S);
    float best = sigmoidf_ref(mp[hm0]);
    int label = 0;
    for (int c = 1; c < S.ncls; ++c) {
      const float v = sigmoidf_ref(mp[hm0 + c]);
      if (v > best) { best = v; label = c; }
    }
    float* o = cand_boxes + ((size_t)set * pre_cap + i) * box_dim;
    o[0] = x; o[1] = y; o[2] = z; o[3] = d0; o[4] = d1; o[5] = d2;
    if (box_dim == 9) {
      o[6] = S.has_vel ? mp[ch_vel(S)] : 0.f;
      o[7] = S.has_vel ? mp[ch_vel(S) + 1] : 0.f;
      o[8] = rot;
    } else {
      o[6] = rot;
    }
    // to_pcdet (iou3d_nms_utils.py:30-34): (x, y, z, dy, dx, dz, -rot - pi/2)
    float* q = nms_boxes + ((size_t)set * pre_cap + i) * 7;
    q[0] = x; q[1] = y; q[2] = z; q[3] = d1; q[4] = d0; q[5] = d2;
    q[6] = -rot - 1.57079632679489661923f;
    cand_scores[(size_t)set * pre_cap + i] = S.use_rectify ? __uint_as_float((unsigned)(e >> 32)) : best;
    cand_labels[(size_t)set * pre_cap + i] = label + S.label_offset;
  }
}

// Rulebooks for evaluating the regression heads at the selected cells only, one block of rows per set.
// The rows of a set are the DISTINCT cells of the 3x3 neighbourhoods of its selected cells (ranked cells come
// in clusters, so this is a few times fewer than nine per cell), in ascending cell order: a bit per map cell
// in LDS, per-word ranks by a popcount scan.  The tower conv gathers each such cell's own 3x3 neighbourhood
// from the shared map (nbr_tower [9][cap9], cap9 = 9*pre_cap), the output conv of selected cell i sums the
// tower rows of its nine neighbours (nbr_out [9][pre_cap]); a neighbour outside the map is zero padding of
// the OUTPUT conv, hence -1 there.  row_counts[set] = rows in use (the caller bounds its GEMMs with it).
static constexpr int LZ_MAX_WORDS = 4096;   // maps of up to 262 144 cells
__global__ __launch_bounds__(SEL_THREADS) void lazy_rules_kernel(DecodeGroups G, int B, int pre_cap,
                                                                 const int32_t* __restrict__ sel_cells,
                                                                 const int32_t* __restrict__ counts,
                                                                 int32_t* __restrict__ nbr_tower,
                                                                 int32_t* __restrict__ nbr_out,
                                                                 int32_t* __restrict__ row_counts) {
  __shared__ unsigned long long bits[LZ_MAX_WORDS];
  __shared__ int rank[LZ_MAX_WORDS];
  __shared__ int wsum[SEL_THREADS / 64];
  const int set = blockIdx.x, group = set / B, b = set - group * B;
  const DecodeSetDev& S = G.g[group];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cells = S.H * S.W, words = (cells + 63) >> 6;
  const int n = counts[set];
  const long long R = (long long)pre_cap * 9;
  int32_t* nt = nbr_tower + (long long)set * 9 * R;
  int32_t* no = nbr_out + (long long)set * 9 * pre_cap;
  for (int w = tid; w < words; w += SEL_THREADS) bits[w] = 0ull;
  __syncthreads();
  for (int i = tid; i < n; i += SEL_THREADS) {
    const int cell = sel_cells[(size_t)set * pre_cap + i];
    const int y = cell / S.W, x = cell - y * S.W;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int ny = y + k / 3 - 1, nx = x + k % 3 - 1;
      if (ny >= 0 && ny < S.H && nx >= 0 && nx < S.W) {
        const int c = ny * S.W + nx;
        atomicOr(&bits[c >> 6], 1ull << (c & 63));
      }
    }
  }
  __syncthreads();
  // exclusive scan of the word popcounts: thread t owns words 4t .. 4t+3
  constexpr int WPT = LZ_MAX_WORDS / SEL_THREADS;
  int pc[WPT], mine = 0;
#pragma unroll
  for (int j = 0; j < WPT; ++j) {
    const int w = tid * WPT + j;
    pc[j] = w < words ? __popcll(bits[w]) : 0;
    mine += pc[j];
  }
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d);
    if (lane >= d) incl += v;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int before = 0, total = 0;
  for (int w = 0; w < SEL_THREADS / 64; ++w) {
    const int c = wsum[w];
    before += w < wv ? c : 0;
    total += c;
  }
  int run = before + incl - mine;
#pragma unroll
  for (int j = 0; j < WPT; ++j) {
    const int w = tid * WPT + j;
    if (w < words) rank[w] = run;
    run += pc[j];
  }
  if (tid == 0) row_counts[set] = total;
  __syncthreads();
  for (int i = tid; i < pre_cap; i += SEL_THREADS) {
    const bool valid = i < n;
    const int cell = valid ? sel_cells[(size_t)set * pre_cap + i] : 0;
    const int y = cell / S.W, x = cell - y * S.W;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int ny = y + k / 3 - 1, nx = x + k % 3 - 1;
      int row = -1;
      if (valid && ny >= 0 && ny < S.H && nx >= 0 && nx < S.W) {
        const int c = ny * S.W + nx;
        row = rank[c >> 6] + __popcll(bits[c >> 6] & ((1ull << (c & 63)) - 1ull));
      }
      no[(long long)k * pre_cap + i] = row;
    }
  }
  // row -> cell: the centre tap's own entry (a row's cell is its centre-tap source), written word by word;
  // the other eight taps are then filled row-parallel with coalesced stores
  const int32_t cell0 = (int32_t)((long long)b * cells);
  for (int w = tid; w < words; w += SEL_THREADS) {
    unsigned long long m = bits[w];
    int row = rank[w];
    while (m) {
      const int bit = __builtin_ctzll(m);
      m &= m - 1ull;
      nt[4 * R + row] = cell0 + (w << 6) + bit;
      ++row;
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int row = tid; row < total; row += SEL_THREADS) {
    const int c = nt[4 * R + row] - cell0;
    const int y = c / S.W, x = c - y * S.W;
#pragma unroll
    for (int k2 = 0; k2 < 9; ++k2) {
      if (k2 == 4) continue;
      const int my = y + k2 / 3 - 1, mx = x + k2 % 3 - 1;
      const bool in2 = my >= 0 && my < S.H && mx >= 0 && mx < S.W;
      nt[(long long)k2 * R + row] = in2 ? cell0 + my * S.W + mx : -1;
    }
  }
}

// Box decode of the selected cells from their lazily evaluated regression rows (reg, height, dim, rot, [vel],
// [iou]; row (b*pre_cap + i) of the group's block), the centre-range mask that decode_score could not apply
// without them, and a stable compaction of the survivors (score order is kept).  A selected cell that fails
// the mask makes the set one candidate short of what the reference would rank; when the set had more
// candidates than pre_max the host must redo the frame with the dense head (status bit 1).
__global__ __launch_bounds__(SEL_THREADS) void decode_lazy_kernel(
    DecodeGroups G, DecodeCommonDev C, int pre_cap, int box_dim, const int32_t* __restrict__ sel_cells,
    const int32_t* __restrict__ totals, float* __restrict__ nms_boxes, float* __restrict__ cand_boxes,
    float* __restrict__ cand_scores, int32_t* __restrict__ cand_labels, int32_t* __restrict__ counts,
    int32_t* __restrict__ status) {
  __shared__ int wsum[SEL_THREADS / 64];
  __shared__ int s_base;
  const int set = blockIdx.x, group = set / C.B, b = set - group * C.B;
  const DecodeSetDev& S = G.g[group];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = counts[set];
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += SEL_THREADS) {
    const int i = i0 + tid;
    bool ok = false;
    float x = 0, y = 0, z = 0, d0 = 0, d1 = 0, d2 = 0, rot = 0, v0 = 0, v1 = 0, score = 0;
    int label = 0;
    if (i < n) {
      const int cell = sel_cells[(size_t)set * pre_cap + i];
      const float* mp = S.maps + ((size_t)b * pre_cap + i) * S.LD;
      const int yy = cell / S.W, xx = cell - yy * S.W;
      x = centre_coord(xx, mp[0], S.stride, C.pillar_size, C.x_min);
      y = centre_coord(yy, mp[1], S.stride, C.pillar_size, C.y_min);
      z = mp[2];
      d0 = expf(fminf(fmaxf(mp[3], -1.2f), 3.2f));
      d1 = expf(fminf(fmaxf(mp[4], -1.2f), 3.2f));
      d2 = expf(fminf(fmaxf(mp[5], -1.2f), 3.2f));
      rot = atan2f(mp[6], mp[7]);
      if (S.has_vel) { v0 = mp[ch_vel(S)]; v1 = mp[ch_vel(S) + 1]; }
      score = cand_scores[(size_t)set * pre_cap + i];
      label = cand_labels[(size_t)set * pre_cap + i];
      ok = !C.has_range || (x >= C.range[0] && y >= C.range[1] && z >= C.range[2] && x <= C.range[3] &&
                            y <= C.range[4] && z <= C.range[5]);
    }
    const unsigned long long m = __ballot(ok);
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();   // also: every thread has read its score / label before anyone overwrites a slot
    int before = 0, total = 0;
    for (int w = 0; w < SEL_THREADS / 64; ++w) {
      const int c = wsum[w];
      before += w < wv ? c : 0;
      total += c;
    }
    const int base = s_base;
    if (ok) {
      const int pos = base + before + __popcll(m & ((1ull << lane) - 1ull));   // pos <= i: slots behind are done
      float* o = cand_boxes + ((size_t)set * pre_cap + pos) * box_dim;
      o[0] = x; o[1] = y; o[2] = z; o[3] = d0; o[4] = d1; o[5] = d2;
      if (box_dim == 9) { o[6] = v0; o[7] = v1; o[8] = rot; } else { o[6] = rot; }
      float* q = nms_boxes + ((size_t)set * pre_cap + pos) * 7;
      q[0] = x; q[1] = y; q[2] = z; q[3] = d1; q[4] = d0; q[5] = d2;
      q[6] = -rot - 1.57079632679489661923f;
      cand_scores[(size_t)set * pre_cap + pos] = score;
      cand_labels[(size_t)set * pre_cap + pos] = label;
    }
    __syncthreads();
    if (tid == 0) s_base = base + total;
    __syncthreads();
  }
  if (tid == 0) {
    const int kept = s_base;
    counts[set] = kept;
    if (kept < n && totals[set] > n) atomicOr(&status[0], 2);
  }
}

__global__ __launch_bounds__(256) void gather_detections_kernel(
    int B, int num_groups, int pre_cap, int box_dim, int keep_stride, int out_cap,
    const float* __restrict__ cand_boxes, const float* __restrict__ cand_scores,
    const int32_t* __restrict__ cand_labels, const int32_t* __restrict__ keep,
    const int32_t* __restrict__ num_keep, float* __restrict__ out_boxes,
    float* __restrict__ out_scores, int64_t* __restrict__ out_labels, int32_t* __restrict__ out_num,
    const int32_t* __restrict__ status_src, int32_t* __restrict__ status_dst) {
  // one block per (group, sample): every block sums the kept counts of the earlier groups itself (all loads
  // independent), so the groups copy in parallel instead of one after the other
  const int g = blockIdx.x, b = blockIdx.y;
  int base = 0, total = 0;
  for (int q = 0; q < num_groups; ++q) {
    const int c = num_keep[q * B + b];
    base += q < g ? c : 0;
    total += c;
  }
  const int set = g * B + b;
  const int nk = num_keep[set];
  for (int i = threadIdx.x; i < nk; i += 256) {
    const int src = keep[(size_t)set * keep_stride + i];
    const int dst = base + i;
    if (dst >= out_cap) continue;
    const float* sb = cand_boxes + ((size_t)set * pre_cap + src) * box_dim;
    float* ob = out_boxes + ((size_t)b * out_cap + dst) * box_dim;
    for (int d = 0; d < box_dim; ++d) ob[d] = sb[d];
    out_scores[(size_t)b * out_cap + dst] = cand_scores[(size_t)set * pre_cap + src];
    out_labels[(size_t)b * out_cap + dst] = cand_labels[(size_t)set * pre_cap + src];
  }
  if (g == 0 && threadIdx.x == 0) out_num[b] = total < out_cap ? total : out_cap;
  if (g == 0 && b == 0 && threadIdx.x == 0 && status_dst) status_dst[0] = status_src[0];   // (no 4-byte copy launch)
}

// ------------------------------------------------------------------------------------------------
extern "C" int pn_nms_rotated(const float* boxes, int num_sets, int cap, const int32_t* counts_dev,
                              int count_host, const float* thresh_dev, float thresh,
                              const int32_t* post_max_dev, int post_max, int keep_stride,
                              int32_t* keep, int32_t* num_keep, void* workspace,
                              size_t workspace_bytes, pnStream stream);
extern "C" size_t pn_nms_workspace_bytes(int num_sets, int cap);

struct DecodeLayout {
  size_t keys, hist, nms_boxes, cand_boxes, cand_scores, cand_labels, counts, keep, num_keep,
      thr, post, status, sel_cells, totals, nms_ws, total;
};

static DecodeLayout decode_layout(const pnDecodeGroup* groups, int num_groups, int B, int pre_cap,
                                  int box_dim) {
  DecodeLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += pn_align_up(bytes, 256); return o; };
  long long cells = 0;
  for (int g = 0; g < num_groups; ++g) cells += (long long)groups[g].height * groups[g].width * B;
  const int S = num_groups * B;
  L.hist = take((size_t)S * DEC_BINS * 4);       // hist, status, counts contiguous: one memset
  L.status = take(256);
  L.keys = take((size_t)cells * 4);
  L.nms_boxes = take((size_t)S * pre_cap * 7 * 4);
  L.cand_boxes = take((size_t)S * pre_cap * box_dim * 4);
  L.cand_scores = take((size_t)S * pre_cap * 4);
  L.cand_labels = take((size_t)S * pre_cap * 4);
  L.counts = take((size_t)S * 4);
  L.keep = take((size_t)S * pre_cap * 4);
  L.num_keep = take((size_t)S * 4);
  L.thr = take((size_t)S * 4);
  L.post = take((size_t)S * 4);
  L.sel_cells = take((size_t)S * pre_cap * 4);
  L.totals = take((size_t)S * 4);
  L.nms_ws = take(pn_nms_workspace_bytes(S, pre_cap));
  L.total = off;
  return L;
}

static int decode_pre_cap(const pnDecodeGroup* groups, int num_groups) {
  int cap = 1;
  for (int g = 0; g < num_groups; ++g) {
    int p = groups[g].pre_max;
    const long long cells = (long long)groups[g].height * groups[g].width;
    if (p > cells) p = (int)cells;
    if (p > cap) cap = p;
  }
  return cap;
}

extern "C" size_t pn_decode_workspace_bytes(const pnDecodeGroup* groups, int num_groups, int batch,
                                            int box_dim) {
  if (!groups || num_groups < 1 || num_groups > PN_MAX_GROUPS || batch < 1) return 0;
  return decode_layout(groups, num_groups, batch, decode_pre_cap(groups, num_groups), box_dim).total;
}

// Everything the three entry points share: argument checks, the per-group device descriptors, the key
// histogram geometry and the workspace layout.  kind 0: full maps (reg .. hm); 1: heat-map channels only;
// 2: regression rows of the selected cells (reg .. [iou], no hm).
struct DecodeSetup {
  DecodeGroups G;
  DecodeCommonDev C;
  DecodeLayout L;
  int S, pre_cap, max_cells;
};

static int decode_setup(const char* who, const pnDecodeGroup* groups, int num_groups, const pnDecodeCommon* common,
                        int box_dim, void* workspace, size_t workspace_bytes, int kind, DecodeSetup& st) {
  PN_CHECK_ARG(groups && common && workspace, "%s: null argument", who);
  PN_CHECK_ARG(num_groups >= 1 && num_groups <= PN_MAX_GROUPS, "%s: %d groups (max %d)", who, num_groups,
               PN_MAX_GROUPS);
  PN_CHECK_ARG(box_dim == 7 || box_dim == 9, "%s: box_dim %d", who, box_dim);
  const int B = common->batch;
  PN_CHECK_ARG(B >= 1 && B <= PN_MAX_BATCH, "%s: batch %d not in 1..%d", who, B, PN_MAX_BATCH);
  st.pre_cap = decode_pre_cap(groups, num_groups);
  PN_CHECK_ARG(st.pre_cap <= 4096, "%s: nms_pre_max_size %d > 4096", who, st.pre_cap);
  st.L = decode_layout(groups, num_groups, B, st.pre_cap, box_dim);
  if (workspace_bytes < st.L.total) {
    pn_set_error("%s: workspace %zu < %zu bytes", who, workspace_bytes, st.L.total);
    return PN_ENOSPC;
  }
  DecodeCommonDev& C = st.C;
  C.B = B; C.num_groups = num_groups;
  C.pillar_size = common->pillar_size; C.x_min = common->x_min; C.y_min = common->y_min;
  C.has_range = common->has_range;
  for (int i = 0; i < 6; ++i) C.range[i] = common->range[i];
  float min_thr = 1e30f;
  long long key_off = 0;
  st.max_cells = 1;
  st.S = num_groups * B;
  for (int g = 0; g < num_groups; ++g) {
    const pnDecodeGroup& s = groups[g];
    const int want = kind == 1 ? s.num_classes
                               : 8 + 2 * s.has_vel + s.has_iou + (kind == 0 ? s.num_classes : 0);
    PN_CHECK_ARG(s.maps && s.height > 0 && s.width > 0 && s.num_classes >= 1 && s.channels == want &&
                     (s.row_stride == 0 || s.row_stride >= s.channels),
                 "%s: group %d channel layout mismatch (%d channels, expected %d)", who, g, s.channels, want);
    PN_CHECK_ARG(s.pre_max >= 1 && s.post_max >= 1, "%s: group %d bad pre/post", who, g);
    PN_CHECK_ARG(kind == 0 || s.rectifier == 0.f, "%s: group %d: a rectified score needs the dense maps", who, g);
    DecodeSetDev& d = st.G.g[g];
    d.maps = s.maps; d.H = s.height; d.W = s.width; d.CH = s.channels; d.ncls = s.num_classes;
    d.LD = s.row_stride > 0 ? s.row_stride : s.channels;
    d.has_vel = s.has_vel; d.has_iou = s.has_iou; d.class_filter = s.class_filter;
    d.label_offset = s.label_offset; d.stride = s.stride; d.score_thr = s.score_threshold;
    d.rectifier = s.rectifier; d.nms_thr = s.nms_threshold; d.use_rectify = s.use_rectify;
    d.pre_max = s.pre_max; d.post_max = s.post_max; d.key_offset = key_off;
    d.hm_only = kind == 1;
    key_off += (long long)s.height * s.width * B;
    if (s.height * s.width > st.max_cells) st.max_cells = s.height * s.width;
    // histogram origin: the smallest rectified score that can pass.  With a rectifier the
    // rectified score may fall below the raw threshold, so start at 0 then.
    const float lo = s.rectifier != 0.f ? 0.f : s.score_threshold;
    if (lo < min_thr) min_thr = lo;
  }
  if (!(min_thr > 0.f)) min_thr = 0.f;
  union { float f; unsigned u; } lo, hi;
  lo.f = min_thr; hi.f = 1.0f;
  C.key_lo = lo.u;
  const unsigned span = hi.u - lo.u + 1;
  int shift = 0;
  while (((unsigned long long)span >> shift) > (unsigned long long)DEC_BINS) ++shift;
  C.key_shift = shift;
  return PN_OK;
}

// score + select: keys / histogram, then per set the sorted top pre_max (boxes decoded unless hm_only)
static int decode_score_select(const DecodeSetup& st, int num_groups, int box_dim, char* ws, hipStream_t stream) {
  const DecodeLayout& L = st.L;
  { const int rc_ = pn_fill_async(ws + L.hist, 0, L.keys - L.hist, stream); if (rc_ != PN_OK) return rc_; }  // hist + status
  int bx = pn_div_up(st.max_cells, 256);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(decode_score_kernel, dim3(bx, num_groups, st.C.B), dim3(256), 0, stream, st.G, st.C,
                     (unsigned*)(ws + L.keys), (unsigned*)(ws + L.hist));
  PN_CHECK_LAUNCH("decode_score_kernel");
  hipLaunchKernelGGL(decode_select_kernel, dim3(st.S), dim3(SEL_THREADS), 0, stream, st.G, st.C,
                     (const unsigned*)(ws + L.keys), (const unsigned*)(ws + L.hist), st.pre_cap, box_dim,
                     (float*)(ws + L.nms_boxes), (float*)(ws + L.cand_boxes),
                     (float*)(ws + L.cand_scores), (int32_t*)(ws + L.cand_labels),
                     (int32_t*)(ws + L.counts), (float*)(ws + L.thr), (int32_t*)(ws + L.post),
                     (int32_t*)(ws + L.status), (int32_t*)(ws + L.sel_cells), (int32_t*)(ws + L.totals));
  PN_CHECK_LAUNCH("decode_select_kernel");
  return PN_OK;
}

// NMS on the sorted candidate boxes of every set, then the per-sample concatenation
static int decode_nms_gather(const DecodeSetup& st, int num_groups, int box_dim, int out_cap, float* out_boxes,
                             float* out_scores, int64_t* out_labels, int32_t* out_num, int32_t* status, char* ws,
                             hipStream_t stream) {
  const DecodeLayout& L = st.L;
  const int S = st.S, pre_cap = st.pre_cap, B = st.C.B;
  int rc = pn_nms_rotated((const float*)(ws + L.nms_boxes), S, pre_cap, (const int32_t*)(ws + L.counts),
                          0, (const float*)(ws + L.thr), 0.f, (const int32_t*)(ws + L.post), pre_cap,
                          pre_cap, (int32_t*)(ws + L.keep), (int32_t*)(ws + L.num_keep),
                          ws + L.nms_ws, L.total - L.nms_ws, stream);
  if (rc != PN_OK) return rc;
  hipLaunchKernelGGL(gather_detections_kernel, dim3(num_groups, B), dim3(256), 0, stream, B, num_groups, pre_cap,
                     box_dim, pre_cap, out_cap, (const float*)(ws + L.cand_boxes),
                     (const float*)(ws + L.cand_scores), (const int32_t*)(ws + L.cand_labels),
                     (const int32_t*)(ws + L.keep), (const int32_t*)(ws + L.num_keep), out_boxes,
                     out_scores, out_labels, out_num, (const int32_t*)(ws + L.status), status);
  PN_CHECK_LAUNCH("gather_detections_kernel");
  return PN_OK;
}

extern "C" int pn_centerhead_detect(const pnDecodeGroup* groups, int num_groups,
                                    const pnDecodeCommon* common, int box_dim, int out_cap,
                                    float* out_boxes, float* out_scores, int64_t* out_labels,
                                    int32_t* out_num, int32_t* status, void* workspace,
                                    size_t workspace_bytes, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(out_boxes && out_scores && out_labels && out_num && out_cap >= 1,
               "pn_centerhead_detect: null output / bad out_cap");
  DecodeSetup st;
  int rc = decode_setup("pn_centerhead_detect", groups, num_groups, common, box_dim, workspace, workspace_bytes, 0, st);
  if (rc != PN_OK) return rc;
  char* ws = (char*)workspace;
  rc = decode_score_select(st, num_groups, box_dim, ws, stream);
  if (rc != PN_OK) return rc;
  return decode_nms_gather(st, num_groups, box_dim, out_cap, out_boxes, out_scores, out_labels, out_num, status, ws,
                           stream);
}

extern "C" int pn_centerhead_select(const pnDecodeGroup* groups, int num_groups, const pnDecodeCommon* common,
                                    int box_dim, int32_t* nbr_tower, int32_t* nbr_out, int32_t* row_counts,
                                    void* workspace, size_t workspace_bytes, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(nbr_tower && nbr_out && row_counts, "pn_centerhead_select: null rulebook output");
  for (int g = 0; groups && g < num_groups; ++g)
    PN_CHECK_ARG((long long)groups[g].height * groups[g].width <= 64ll * LZ_MAX_WORDS,
                 "pn_centerhead_select: group %d map of %d x %d cells exceeds %d", g, groups[g].height,
                 groups[g].width, 64 * LZ_MAX_WORDS);
  DecodeSetup st;
  int rc = decode_setup("pn_centerhead_select", groups, num_groups, common, box_dim, workspace, workspace_bytes, 1, st);
  if (rc != PN_OK) return rc;
  char* ws = (char*)workspace;
  rc = decode_score_select(st, num_groups, box_dim, ws, stream);
  if (rc != PN_OK) return rc;
  hipLaunchKernelGGL(lazy_rules_kernel, dim3(st.S), dim3(SEL_THREADS), 0, stream, st.G, st.C.B, st.pre_cap,
                     (const int32_t*)(ws + st.L.sel_cells), (const int32_t*)(ws + st.L.counts), nbr_tower, nbr_out,
                     row_counts);
  PN_CHECK_LAUNCH("lazy_rules_kernel");
  return PN_OK;
}

extern "C" int pn_centerhead_select_capacity(const pnDecodeGroup* groups, int num_groups) {
  if (!groups || num_groups < 1 || num_groups > PN_MAX_GROUPS) return 0;
  return decode_pre_cap(groups, num_groups);
}

extern "C" int pn_centerhead_detect_selected(const pnDecodeGroup* groups, int num_groups,
                                             const pnDecodeCommon* common, int box_dim, int out_cap,
                                             float* out_boxes, float* out_scores, int64_t* out_labels,
                                             int32_t* out_num, int32_t* status, void* workspace,
                                             size_t workspace_bytes, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(out_boxes && out_scores && out_labels && out_num && out_cap >= 1,
               "pn_centerhead_detect_selected: null output / bad out_cap");
  DecodeSetup st;
  int rc = decode_setup("pn_centerhead_detect_selected", groups, num_groups, common, box_dim, workspace,
                        workspace_bytes, 2, st);
  if (rc != PN_OK) return rc;
  char* ws = (char*)workspace;
  const DecodeLayout& L = st.L;
  hipLaunchKernelGGL(decode_lazy_kernel, dim3(st.S), dim3(SEL_THREADS), 0, stream, st.G, st.C, st.pre_cap, box_dim,
                     (const int32_t*)(ws + L.sel_cells), (const int32_t*)(ws + L.totals),
                     (float*)(ws + L.nms_boxes), (float*)(ws + L.cand_boxes), (float*)(ws + L.cand_scores),
                     (int32_t*)(ws + L.cand_labels), (int32_t*)(ws + L.counts), (int32_t*)(ws + L.status));
  PN_CHECK_LAUNCH("decode_lazy_kernel");
  return decode_nms_gather(st, num_groups, box_dim, out_cap, out_boxes, out_scores, out_labels, out_num, status, ws,
                           stream);
}
