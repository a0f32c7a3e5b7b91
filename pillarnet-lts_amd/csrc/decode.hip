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
    const int hm0 = ch_hm(S);
    float best = sigmoidf_ref(m[hm0]);
    int label = 0;
    for (int c = 1; c < S.ncls; ++c) {
      const float v = sigmoidf_ref(m[hm0 + c]);
      if (v > best) { best = v; label = c; }
    }
    bool ok = best > S.score_thr;
    if (S.class_filter >= 0) ok = ok && label == S.class_filter;
    if (ok && C.has_range) {
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

__global__ __launch_bounds__(SEL_THREADS) void decode_select_kernel(
    DecodeGroups G, DecodeCommonDev C, const unsigned* __restrict__ keys,
    const unsigned* __restrict__ hist, int pre_cap, int box_dim, float* __restrict__ nms_boxes,
    float* __restrict__ cand_boxes, float* __restrict__ cand_scores,
    int32_t* __restrict__ cand_labels, int32_t* __restrict__ counts, float* __restrict__ nms_thr,
    int32_t* __restrict__ nms_post, int32_t* __restrict__ status) {
  __shared__ unsigned long long list[DEC_LIST];
  __shared__ int sfx[SEL_THREADS];
  __shared__ int s_cut, s_cnt;
  const int set = blockIdx.x, group = set / C.B, b = set - group * C.B;
  const DecodeSetDev& S = G.g[group];
  const int cells = S.H * S.W, tid = threadIdx.x;
  int pre_max = S.pre_max < pre_cap ? S.pre_max : pre_cap;

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
  if (tid == 0) { s_cut = 0; s_cnt = 0; }
  __syncthreads();
  {
    const int above = sfx[tid] - mine;  // keys in bins above this thread's bins
    if (above < pre_max && sfx[tid] >= pre_max) {
      int run = above;
      for (int k = 0; k < BPT; ++k) {
        const int bin = DEC_BINS - 1 - (tid * BPT + k);
        run += (int)h[bin];
        if (run >= pre_max) { s_cut = bin; break; }
      }
    }
  }
  __syncthreads();
  const int cut = s_cut;  // 0 when the set holds fewer than pre_max candidates

  // ---- collect ----------------------------------------------------------------------------------
  const unsigned* kk = keys + S.key_offset + (long long)b * cells;
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
      const unsigned key = kv[u];
      if (!key) continue;
      const int cell = base + u * SEL_THREADS + tid;
      int bin = key > C.key_lo ? (int)((key - C.key_lo) >> C.key_shift) : 0;
      if (bin >= DEC_BINS) bin = DEC_BINS - 1;
      if (bin >= cut) {
        const int pos = atomicAdd(&s_cnt, 1);
        if (pos < DEC_LIST)
          list[pos] = ((unsigned long long)key << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)cell);
      }
    }
  }
  __syncthreads();
  int n = s_cnt;
  if (n > DEC_LIST) {
    // More cells inside the single cut-off bin than the list holds (maps full of equal scores): report it,
    // and collect again by a fixed rule -- every key above the cut-off bin, then the cut-off bin's cells in
    // ascending cell order until the list is full -- so that the result does not depend on which thread's
    // atomicAdd came first.  Only this degenerate case pays for the ordered pass.
    if (tid == 0) { atomicOr(&status[0], 1); s_cnt = 0; }
    __syncthreads();
    for (int cell = tid; cell < cells; cell += SEL_THREADS) {
      const unsigned key = kk[cell];
      if (!key) continue;
      int bin = key > C.key_lo ? (int)((key - C.key_lo) >> C.key_shift) : 0;
      if (bin >= DEC_BINS) bin = DEC_BINS - 1;
      if (bin > cut) {   // fewer than pre_max of these by the definition of the cut-off bin
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
        int bin = key > C.key_lo ? (int)((key - C.key_lo) >> C.key_shift) : 0;
        if (bin >= DEC_BINS) bin = DEC_BINS - 1;
        if (!(key != 0u && bin == cut)) key = 0u;
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
  int p2 = 1;
  while (p2 < n) p2 <<= 1;
  for (int i = n + tid; i < p2; i += SEL_THREADS) list[i] = 0ull;
  __syncthreads();
  bitonic_sort_desc(list, p2);
  const int m = n < pre_max ? n : pre_max;
  if (tid == 0) {
    counts[set] = m;
    nms_thr[set] = S.nms_thr;
    nms_post[set] = S.post_max < pre_cap ? S.post_max : pre_cap;
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

__global__ __launch_bounds__(256) void gather_detections_kernel(
    int B, int num_groups, int pre_cap, int box_dim, int keep_stride, int out_cap,
    const float* __restrict__ cand_boxes, const float* __restrict__ cand_scores,
    const int32_t* __restrict__ cand_labels, const int32_t* __restrict__ keep,
    const int32_t* __restrict__ num_keep, float* __restrict__ out_boxes,
    float* __restrict__ out_scores, int32_t* __restrict__ out_labels, int32_t* __restrict__ out_num) {
  const int b = blockIdx.x;
  int base = 0;
  for (int g = 0; g < num_groups; ++g) {
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
    base += nk;
  }
  if (threadIdx.x == 0) out_num[b] = base < out_cap ? base : out_cap;
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
      thr, post, status, nms_ws, total;
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

extern "C" int pn_centerhead_detect(const pnDecodeGroup* groups, int num_groups,
                                    const pnDecodeCommon* common, int box_dim, int out_cap,
                                    float* out_boxes, float* out_scores, int32_t* out_labels,
                                    int32_t* out_num, int32_t* status, void* workspace,
                                    size_t workspace_bytes, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(groups && common && out_boxes && out_scores && out_labels && out_num && workspace,
               "pn_centerhead_detect: null argument");
  PN_CHECK_ARG(num_groups >= 1 && num_groups <= PN_MAX_GROUPS, "pn_centerhead_detect: %d groups (max %d)",
               num_groups, PN_MAX_GROUPS);
  PN_CHECK_ARG(box_dim == 7 || box_dim == 9, "pn_centerhead_detect: box_dim %d", box_dim);
  const int B = common->batch;
  PN_CHECK_ARG(B >= 1 && out_cap >= 1, "pn_centerhead_detect: bad batch / out_cap");
  const int pre_cap = decode_pre_cap(groups, num_groups);
  PN_CHECK_ARG(pre_cap <= 4096, "pn_centerhead_detect: nms_pre_max_size %d > 4096", pre_cap);
  const DecodeLayout L = decode_layout(groups, num_groups, B, pre_cap, box_dim);
  if (workspace_bytes < L.total) {
    pn_set_error("pn_centerhead_detect: workspace %zu < %zu bytes", workspace_bytes, L.total);
    return PN_ENOSPC;
  }
  char* ws = (char*)workspace;
  DecodeGroups G;
  DecodeCommonDev C;
  C.B = B; C.num_groups = num_groups;
  C.pillar_size = common->pillar_size; C.x_min = common->x_min; C.y_min = common->y_min;
  C.has_range = common->has_range;
  for (int i = 0; i < 6; ++i) C.range[i] = common->range[i];
  float min_thr = 1e30f;
  long long key_off = 0;
  int max_cells = 1;
  const int S = num_groups * B;
  PN_CHECK_ARG(B <= PN_MAX_BATCH, "pn_centerhead_detect: batch %d > %d", B, PN_MAX_BATCH);
  for (int g = 0; g < num_groups; ++g) {
    const pnDecodeGroup& s = groups[g];
    PN_CHECK_ARG(s.maps && s.height > 0 && s.width > 0 && s.num_classes >= 1 &&
                     s.channels == 8 + 2 * s.has_vel + s.has_iou + s.num_classes &&
                     (s.row_stride == 0 || s.row_stride >= s.channels),
                 "pn_centerhead_detect: group %d channel layout mismatch", g);
    PN_CHECK_ARG(s.pre_max >= 1 && s.post_max >= 1, "pn_centerhead_detect: group %d bad pre/post", g);
    DecodeSetDev& d = G.g[g];
    d.maps = s.maps; d.H = s.height; d.W = s.width; d.CH = s.channels; d.ncls = s.num_classes;
    d.LD = s.row_stride > 0 ? s.row_stride : s.channels;
    d.has_vel = s.has_vel; d.has_iou = s.has_iou; d.class_filter = s.class_filter;
    d.label_offset = s.label_offset; d.stride = s.stride; d.score_thr = s.score_threshold;
    d.rectifier = s.rectifier; d.nms_thr = s.nms_threshold; d.use_rectify = s.use_rectify;
    d.pre_max = s.pre_max; d.post_max = s.post_max; d.key_offset = key_off;
    key_off += (long long)s.height * s.width * B;
    if (s.height * s.width > max_cells) max_cells = s.height * s.width;
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

  PN_CHECK_HIP(hipMemsetAsync(ws + L.hist, 0, L.keys - L.hist, stream));  // hist + status

  int bx = pn_div_up(max_cells, 256);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(decode_score_kernel, dim3(bx, num_groups, B), dim3(256), 0, stream, G, C,
                     (unsigned*)(ws + L.keys), (unsigned*)(ws + L.hist));
  PN_CHECK_LAUNCH("decode_score_kernel");
  hipLaunchKernelGGL(decode_select_kernel, dim3(S), dim3(SEL_THREADS), 0, stream, G, C,
                     (const unsigned*)(ws + L.keys), (const unsigned*)(ws + L.hist), pre_cap, box_dim,
                     (float*)(ws + L.nms_boxes), (float*)(ws + L.cand_boxes),
                     (float*)(ws + L.cand_scores), (int32_t*)(ws + L.cand_labels),
                     (int32_t*)(ws + L.counts), (float*)(ws + L.thr), (int32_t*)(ws + L.post),
                     (int32_t*)(ws + L.status));
  PN_CHECK_LAUNCH("decode_select_kernel");
  int rc = pn_nms_rotated((const float*)(ws + L.nms_boxes), S, pre_cap, (const int32_t*)(ws + L.counts),
                          0, (const float*)(ws + L.thr), 0.f, (const int32_t*)(ws + L.post), pre_cap,
                          pre_cap, (int32_t*)(ws + L.keep), (int32_t*)(ws + L.num_keep),
                          ws + L.nms_ws, L.total - L.nms_ws, stream);
  if (rc != PN_OK) return rc;
  hipLaunchKernelGGL(gather_detections_kernel, dim3(B), dim3(256), 0, stream, B, num_groups, pre_cap,
                     box_dim, pre_cap, out_cap, (const float*)(ws + L.cand_boxes),
                     (const float*)(ws + L.cand_scores), (const int32_t*)(ws + L.cand_labels),
                     (const int32_t*)(ws + L.keep), (const int32_t*)(ws + L.num_keep), out_boxes,
                     out_scores, out_labels, out_num);
  PN_CHECK_LAUNCH("gather_detections_kernel");
  if (status)
    PN_CHECK_HIP(hipMemcpyAsync(status, ws + L.status, 4, hipMemcpyDeviceToDevice, stream));
  return PN_OK;
}
