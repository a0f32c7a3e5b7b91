// (a10/a11) rotated-BEV IoU and greedy NMS for gfx950, selection included, no host round trip.
//
// Reference being replaced:
//   det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:36-234   geometry (box_overlap / iou_bev)
//   det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:280-324  nms_kernel: 64x64 tile per 64-thread block,
//                                                        every lane walks 64 column boxes serially
//   det3d/ops/iou3d_nms/src/iou3d_nms.cpp:113-159        cudaMalloc + blocking D2H of the bit matrix +
//                                                        serial greedy loop on the host
//
// Here:
//   nms_mask_kernel    one 256-thread workgroup per *upper-triangle* 64x64 tile (the greedy pass
//                      never reads the lower triangle).  Per-box data (rotated corners, cos/sin,
//                      bounding radius) is computed once into LDS.  All 4096 pairs get a
//                      bounding-circle test; survivors are compacted into an LDS list so that the
//                      expensive polygon clip runs on dense wavefronts instead of one lane in 64.
//                      A pair that fails the circle test has overlap exactly 0 in the reference
//                      (no edge crossing, no corner within the 1 cm slack), so the bit matrix is
//                      unchanged for any threshold >= 0.
//                      (Three passes on dense wavefronts: circles for all pairs, the separating-axis test
//                      for the survivors, and the pairs that remain are queued.)
//   nms_clip_kernel    the polygon clips of every tile's queue, one pair per lane over the whole chip, vertex
//                      lists in LDS; sets bits in the tiles' words with 64-bit atomic ORs.
//   nms_select_kernel  one wavefront per box set.  Lane c owns removal word c; the 64 rows of a
//                      block are resolved with a ctz walk over the still-alive bits on the scalar unit
//                      (readlane of the diagonal word, fetched one block ahead), then the kept rows' later
//                      words are OR-ed in by lane, eight loads in flight.  Stops as soon as post_max boxes are kept.
//   Measured in the configs[1] frame (6 sets x 1000 boxes): mask 62.2 -> 13.4 us (+ clip 11.0, + counter fill 2.5),
//   select 23.5 -> 16.7 us.
//
// -ffp-contract=off: products and sums round separately, as in the reference's CPU twin
// (iou3d_cpu.cpp) which the oracle is pinned to.
#include "common.h"
#include <mutex>

static constexpr float IOU_EPS = 1e-8f;
static constexpr float IN_BOX_MARGIN = 1e-2f;
static constexpr int MAX_POLY = 24;

struct pt2 {
  float x, y;
};

// everything box_overlap needs about one box, computed once
struct BoxGeom {
  pt2 c[4];            // rotated corners
  float cx, cy;        // centre
  float cneg, sneg;    // cos(-heading), sin(-heading)
  float hx_m, hy_m;    // dx/2 + margin, dy/2 + margin
  float area;          // dx*dy
  float radius;        // half diagonal (+ slack) for the circle pre-test
};

__device__ __forceinline__ float cross3(pt2 p1, pt2 p2, pt2 p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__device__ __forceinline__ pt2 rotate_about(float cx, float cy, float c, float s, float px, float py) {
  pt2 r;
  r.x = (px - cx) * c + (py - cy) * (-s) + cx;
  r.y = (px - cx) * s + (py - cy) * c + cy;
  return r;
}

__device__ __forceinline__ void make_geom(const float* __restrict__ b, BoxGeom& g) {
  const float hx = b[3] / 2, hy = b[4] / 2;
  const float x1 = b[0] - hx, y1 = b[1] - hy, x2 = b[0] + hx, y2 = b[1] + hy;
  const float c = cosf(b[6]), s = sinf(b[6]);
  g.cx = b[0];
  g.cy = b[1];
  g.c[0] = rotate_about(b[0], b[1], c, s, x1, y1);
  g.c[1] = rotate_about(b[0], b[1], c, s, x2, y1);
  g.c[2] = rotate_about(b[0], b[1], c, s, x2, y2);
  g.c[3] = rotate_about(b[0], b[1], c, s, x1, y2);
  g.cneg = cosf(-b[6]);
  g.sneg = sinf(-b[6]);
  g.hx_m = hx + IN_BOX_MARGIN;
  g.hy_m = hy + IN_BOX_MARGIN;
  g.area = b[3] * b[4];
  g.radius = sqrtf(hx * hx + hy * hy) * 1.0001f + 0.02f;
}

__device__ __forceinline__ bool seg_intersection(pt2 p1, pt2 p0, pt2 q1, pt2 q0, pt2& out) {
  // iou3d_nms_kernel.cu:63-92
  const bool bb = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
                  fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
  if (!bb) return false;
  const float s1 = cross3(q0, p1, p0);
  const float s2 = cross3(p1, q1, p0);
  const float s3 = cross3(p0, q1, q0);
  const float s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > IOU_EPS) {
    out.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    out.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    out.x = (b0 * c1 - b1 * c0) / D;
    out.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

__device__ __forceinline__ bool point_in_box(const BoxGeom& g, pt2 p) {
  // iou3d_nms_kernel.cu:51-61
  const float rx = (p.x - g.cx) * g.cneg + (p.y - g.cy) * (-g.sneg);
  const float ry = (p.x - g.cx) * g.sneg + (p.y - g.cy) * g.cneg;
  return fabsf(rx) < g.hx_m && fabsf(ry) < g.hy_m;
}

// Vertex list of the clip: x, y, angle per vertex.  The private arrays live in scratch memory (run-time indices);
// PolyLds is the same list in LDS, vertex-major ([field][vertex][thread], conflict-free).  Measured in the frame:
// LDS lists with 128 clip lanes 69 us against 62 us for the private lists on all 256 lanes -- the kernel is bound
// by the clip's arithmetic (~2 500 instructions per pair), not by where its vertices live; the policy stays a
// template parameter for tuning.
struct PolyPrivate {
  float px[MAX_POLY], py[MAX_POLY], pa[MAX_POLY];
  __device__ __forceinline__ float& x(int k) { return px[k]; }
  __device__ __forceinline__ float& y(int k) { return py[k]; }
  __device__ __forceinline__ float& a(int k) { return pa[k]; }
};
template <int NT>
struct PolyLds {
  float* base;   // [3][MAX_POLY][NT], already offset by the thread
  __device__ __forceinline__ float& x(int k) { return base[k * NT]; }
  __device__ __forceinline__ float& y(int k) { return base[(MAX_POLY + k) * NT]; }
  __device__ __forceinline__ float& a(int k) { return base[(2 * MAX_POLY + k) * NT]; }
};

template <typename Poly>
__device__ float box_overlap(const BoxGeom& A, const BoxGeom& B, Poly poly) {
  // iou3d_nms_kernel.cu:104-225
  float sx = 0.f, sy = 0.f;
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pt2 x;
      if (seg_intersection(A.c[(i + 1) & 3], A.c[i], B.c[(j + 1) & 3], B.c[j], x)) {
        sx = sx + x.x;
        sy = sy + x.y;
        poly.x(cnt) = x.x;
        poly.y(cnt) = x.y;
        ++cnt;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (point_in_box(A, B.c[k])) {
      sx = sx + B.c[k].x; sy = sy + B.c[k].y;
      poly.x(cnt) = B.c[k].x;
      poly.y(cnt) = B.c[k].y;
      ++cnt;
    }
    if (point_in_box(B, A.c[k])) {
      sx = sx + A.c[k].x; sy = sy + A.c[k].y;
      poly.x(cnt) = A.c[k].x;
      poly.y(cnt) = A.c[k].y;
      ++cnt;
    }
  }
  if (cnt < 3) return 0.f;  // reference: the area loop adds nothing meaningful (0, or one zero cross)
  const float mx = sx / cnt, my = sy / cnt;
  for (int k = 0; k < cnt; ++k) poly.a(k) = atan2f(poly.y(k) - my, poly.x(k) - mx);
  // same bubble sort as the reference (stable, strict >); angles are evaluated once per vertex
  for (int j = 0; j < cnt - 1; ++j) {
    float a0 = poly.a(0), x0 = poly.x(0), y0 = poly.y(0);   // the element bubbling upwards stays in registers
    for (int i = 0; i < cnt - j - 1; ++i) {
      const float a1 = poly.a(i + 1), x1 = poly.x(i + 1), y1 = poly.y(i + 1);
      if (a0 > a1) {
        poly.a(i) = a1; poly.x(i) = x1; poly.y(i) = y1;
      } else {
        poly.a(i) = a0; poly.x(i) = x0; poly.y(i) = y0;
        a0 = a1; x0 = x1; y0 = y1;
      }
    }
    const int last = cnt - j - 1;
    poly.a(last) = a0; poly.x(last) = x0; poly.y(last) = y0;
  }
  float area = 0.f;
  const float bx = poly.x(0), by = poly.y(0);
  float px = bx, py = by;               // (the k = 0 term is kept: it is 0 for finite vertices, NaN otherwise)
  for (int k = 0; k < cnt - 1; ++k) {
    const float qx = poly.x(k + 1), qy = poly.y(k + 1);
    const float ux = px - bx, uy = py - by;
    const float vx = qx - bx, vy = qy - by;
    area += ux * vy - uy * vx;
    px = qx; py = qy;
  }
  return fabsf(area) / 2.0f;
}
__device__ __forceinline__ float box_overlap(const BoxGeom& A, const BoxGeom& B) {
  PolyPrivate p;
  return box_overlap<PolyPrivate&>(A, B, p);
}

__device__ __forceinline__ float iou_from_overlap(const BoxGeom& A, const BoxGeom& B, float ov) {
  return ov / fmaxf(A.area + B.area - ov, IOU_EPS);
}

__device__ __forceinline__ bool circles_apart(const BoxGeom& A, const BoxGeom& B) {
  const float dx = A.cx - B.cx, dy = A.cy - B.cy;
  const float r = A.radius + B.radius;
  return dx * dx + dy * dy > r * r;  // false for NaN: falls through to the exact path
}

// Separating-axis test in the frames of both boxes, with 5 cm of slack.  Two rectangles that an axis of one of
// them separates by more than the slack share no edge crossing, and no corner of either comes within the 1 cm
// margin of the other (the margin region reaches 1.5 cm beyond a box): the reference's vertex list is empty and
// its overlap exactly 0.  (NaNs compare false: the pair goes to the exact path.)
__device__ __forceinline__ bool axis_separated(const BoxGeom& A, const BoxGeom& B) {
  constexpr float SLACK = 0.05f;
  float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dx = B.c[k].x - A.cx, dy = B.c[k].y - A.cy;
    const float rx = dx * A.cneg - dy * A.sneg, ry = dx * A.sneg + dy * A.cneg;
    xmin = fminf(xmin, rx); xmax = fmaxf(xmax, rx);
    ymin = fminf(ymin, ry); ymax = fmaxf(ymax, ry);
  }
  const float hx = A.hx_m + SLACK, hy = A.hy_m + SLACK;
  return xmin > hx || xmax < -hx || ymin > hy || ymax < -hy;
}

// LDS copy of a box: 17 dwords, so that lanes reading the same field of different boxes (the compacted pair lists
// pair lanes with arbitrary boxes) hit different banks; at 16 dwords all 64 boxes share 4 banks (PMC: 86 % of the
// mask kernel's LDS cycles were conflicts)
struct BoxGeomLds : BoxGeom {
  float pad;
};
static_assert(sizeof(BoxGeom) == 64 && sizeof(BoxGeomLds) == 68, "box geometry layout");

struct NmsPair {
  int32_t row;   // set * cap + position of the (higher-scoring) row box, or -1
  int32_t col;   // position of the column box inside the set
};
static constexpr int NMS_QUEUE_CAP = 1 << 18;

// ------------------------------------------------------------------------------------------------
static constexpr int NMS_TILE = 64;
static constexpr int NMS_THREADS = 256;

__global__ __launch_bounds__(NMS_THREADS) void nms_mask_kernel(
    const float* __restrict__ boxes, const int32_t* __restrict__ counts, int count_is_host_n,
    int cap, int col_blocks, int tiles_per_set, const float* __restrict__ thresh_per_set,
    float thresh_all, unsigned long long* __restrict__ mask, BoxGeom* __restrict__ geoms, int* __restrict__ qcount,
    NmsPair* __restrict__ queue, int qcap) {
  __shared__ BoxGeomLds rowg[NMS_TILE];
  __shared__ BoxGeomLds colg[NMS_TILE];
  __shared__ unsigned long long lmask[NMS_TILE];
  __shared__ unsigned short plist[NMS_TILE * NMS_TILE];
  __shared__ unsigned short plist2[NMS_TILE * NMS_TILE];
  __shared__ int pcount, pcount2, qbase;

  const int set = blockIdx.x / tiles_per_set;
  int tile = blockIdx.x - set * tiles_per_set;
  // unrank the upper-triangle tile: rows rb, columns cb >= rb
  int rb = 0, rem = col_blocks;
  while (tile >= rem) { tile -= rem; ++rb; --rem; }
  const int cb = rb + tile;

  int n = count_is_host_n ? count_is_host_n : counts[set];
  if (n > cap) n = cap;
  if (rb * NMS_TILE >= n || cb * NMS_TILE >= n) return;
  const float thr = thresh_per_set ? thresh_per_set[set] : thresh_all;
  const float* B = boxes + (size_t)set * cap * 7;
  const int tid = threadIdx.x;

  if (tid < NMS_TILE) {
    const int r = rb * NMS_TILE + tid;
    if (r < n) make_geom(B + (size_t)r * 7, rowg[tid]);
    lmask[tid] = 0ull;
  } else if (tid < 2 * NMS_TILE) {
    const int c = cb * NMS_TILE + (tid - NMS_TILE);
    if (c < n) make_geom(B + (size_t)c * 7, colg[tid - NMS_TILE]);
  }
  if (tid == 0) { pcount = 0; pcount2 = 0; }
  __syncthreads();

#ifdef PN_NMS_PROBE_GEOM_ONLY   // tuning builds (wrong results)
  if (thr < 1e30f) return;
#endif
  // Three passes, each on dense wavefronts: bounding circles for all 4096 pairs -> list 1; the separating-axis
  // test for list 1 -> list 2; the polygon clip for list 2.  (Run on all pairs the axis test costs more than it
  // saves: 100 instructions x 4096 pairs against ~2 500 x the few hundred pairs that reach the clip.)
  {
    const int r = tid & 63, c0 = (tid >> 6) * 16;
    const int gr = rb * NMS_TILE + r;
    for (int k = 0; k < 16; ++k) {          // (trip count uniform: the ballots below need every lane)
      const int c = c0 + k, gc = cb * NMS_TILE + c;
      bool close = false;
      if (gr < n && gc < n && gc > gr) {
        if (thr < 0.f) {   // circle NMS (circle_nms_jit.py:5-28): squared centre distance <= -thr suppresses
          const float dx = rowg[r].cx - colg[c].cx, dy = rowg[r].cy - colg[c].cy;
          if (dx * dx + dy * dy <= -thr) atomicOr(&lmask[r], 1ull << c);
        } else {
          close = !circles_apart(rowg[r], colg[c]);
        }
      }
      // one LDS atomic per wavefront and step instead of one per surviving pair (all on one address)
      const unsigned long long votes = __ballot(close);
      if (votes) {
        int base = 0;
        if ((tid & 63) == 0) base = atomicAdd(&pcount, __popcll(votes));
        base = __shfl(base, 0);
        if (close) plist[base + __popcll(votes & ((1ull << (tid & 63)) - 1ull))] = (unsigned short)((r << 6) | c);
      }
    }
  }
  __syncthreads();
  {
    const int n1 = pcount;
    for (int p0 = 0; p0 < n1; p0 += NMS_THREADS) {   // uniform trip count
      const int p = p0 + tid;
      bool hit = false;
      unsigned short rc = 0;
      if (p < n1) {
        rc = plist[p];
        const int r = rc >> 6, c = rc & 63;
        hit = !axis_separated(rowg[r], colg[c]) && !axis_separated(colg[c], rowg[r]);
      }
      const unsigned long long votes = __ballot(hit);
      if (votes) {
        int base = 0;
        if ((tid & 63) == 0) base = atomicAdd(&pcount2, __popcll(votes));
        base = __shfl(base, 0);
        if (hit) plist2[base + __popcll(votes & ((1ull << (tid & 63)) - 1ull))] = rc;
      }
    }
  }
  __syncthreads();
  const int np = pcount2;
  // Hand the clip over to nms_clip_kernel when a queue is given: the pairs that need it are spread very unevenly
  // over the tiles (PMC: the average wavefront of this kernel lives 8.7 k cycles, the launch 100 k -- a few tiles
  // with a thousand overlapping pairs clip for ten rounds while the chip idles), the queue spreads them over
  // every CU.  A tile whose pairs do not fit any more clips them itself, as without a queue.
  bool local_clip = true;
  if (queue) {
    if (cb == rb && tid < NMS_TILE && rb * NMS_TILE + tid < n)
      geoms[(size_t)set * cap + rb * NMS_TILE + tid] = rowg[tid];      // every row block has its diagonal tile
    if (tid == 0) qbase = np ? atomicAdd(qcount, np) : 0;
    __syncthreads();
    const int base = qbase;
    local_clip = base + np > qcap;
    for (int p = tid; p < np && base + p < qcap; p += NMS_THREADS) {
      NmsPair e;
      e.row = local_clip ? -1 : set * cap + rb * NMS_TILE + (plist2[p] >> 6);   // -1: slot reserved, not used
      e.col = cb * NMS_TILE + (plist2[p] & 63);
      queue[base + p] = e;
    }
  }
#ifdef PN_NMS_PROBE_NO_CLIP   // tuning builds: the kernel without its exact phase (wrong results)
  if (thr < 1e30f) return;
#endif
  if (local_clip)
  for (int p = tid; p < np; p += NMS_THREADS) {
    const int r = plist2[p] >> 6, c = plist2[p] & 63;
    const float ov = box_overlap(rowg[r], colg[c]);
    if (iou_from_overlap(rowg[r], colg[c], ov) > thr) atomicOr(&lmask[r], 1ull << c);
  }
  __syncthreads();
  if (tid < NMS_TILE) {
    const int r = rb * NMS_TILE + tid;
    if (r < n)
      mask[((size_t)set * cap + r) * col_blocks + cb] = lmask[tid];
  }
}

// the polygon clips nms_mask_kernel queued, one pair per lane over the whole chip
__global__ __launch_bounds__(64) void nms_clip_kernel(const BoxGeom* __restrict__ geoms, const NmsPair* __restrict__ queue,
                                                      const int* __restrict__ qcount, int qcap, int cap, int col_blocks,
                                                      const float* __restrict__ thresh_per_set, float thresh_all,
                                                      unsigned long long* __restrict__ mask) {
  __shared__ float polys[3 * MAX_POLY * 64];   // vertex lists in LDS: one pair per lane, nothing else competes for it
  PolyLds<64> poly{polys + threadIdx.x};
  int total = *qcount;
  if (total > qcap) total = qcap;
  for (int p = blockIdx.x * 64 + threadIdx.x; p < total; p += gridDim.x * 64) {
    const NmsPair e = queue[p];
    if (e.row < 0) continue;
    const int set = e.row / cap;
    const BoxGeom A = geoms[e.row], B = geoms[(size_t)set * cap + e.col];
    const float thr = thresh_per_set ? thresh_per_set[set] : thresh_all;
    const float ov = box_overlap(A, B, poly);
    if (iou_from_overlap(A, B, ov) > thr)
      atomicOr(&mask[(size_t)e.row * col_blocks + (e.col >> 6)], 1ull << (e.col & 63));
  }
}

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}
// NW = 64-bit removal words per lane: one wave covers 4096 * NW boxes
template <int NW>
__global__ __launch_bounds__(64) void nms_select_kernel(
    const unsigned long long* __restrict__ mask, const int32_t* __restrict__ counts,
    int count_is_host_n, int cap, int col_blocks, const int32_t* __restrict__ post_max_per_set,
    int post_max_all, int keep_stride, int32_t* __restrict__ keep, int32_t* __restrict__ num_keep) {
  const int set = blockIdx.x, lane = threadIdx.x;
  int n = count_is_host_n ? count_is_host_n : counts[set];
  if (n > cap) n = cap;
  if (n < 0) n = 0;
  int post_max = post_max_per_set ? post_max_per_set[set] : post_max_all;
  if (post_max > keep_stride) post_max = keep_stride;
  const int nb = (n + 63) >> 6;
  const unsigned long long* M = mask + (size_t)set * cap * col_blocks;
  int32_t* K = keep + (size_t)set * keep_stride;

  unsigned long long remv[NW];  // lane c, word w: removal bits of block w * 64 + c
#pragma unroll
  for (int w = 0; w < NW; ++w) remv[w] = 0ull;
  int kept_total = 0;
  // the diagonal word of a block depends on nothing but the block: it is requested one block ahead
  unsigned long long diag_next = lane < n ? M[(size_t)lane * col_blocks] : 0ull;
  for (int b = 0; b < nb && kept_total < post_max; ++b) {
    const int row = b * 64 + lane;
    const unsigned long long diag = diag_next;
    {
      const int nrow = row + 64;
      diag_next = (b + 1 < nb && nrow < n) ? M[(size_t)nrow * col_blocks + b + 1] : 0ull;
    }
    unsigned long long mine = remv[0];
#pragma unroll
    for (int w = 1; w < NW; ++w)
      if ((b >> 6) == w) mine = remv[w];          // wave-uniform pick (no dynamic register index)
    unsigned long long removed = readlane64(mine, b & 63);
    const int rows_here = n - b * 64;
    if (rows_here < 64) removed |= ~0ull << rows_here;
    // the walk runs on the scalar unit: the alive mask is wave-uniform
    unsigned long long kept = 0ull, alive = uniform64(~removed);
    while (alive) {
      const int j = __builtin_ctzll(alive);
      kept |= 1ull << j;
      removed |= readlane64(diag, j);
      alive = j == 63 ? 0ull : uniform64(~removed & (~0ull << (j + 1)));
    }
    if ((kept >> lane) & 1ull) {
      const int pos = kept_total + __popcll(kept & ((1ull << lane) - 1ull));
      if (pos < post_max) K[pos] = row;
    }
    kept_total += __popcll(kept);
    // later words of the rows just kept, eight loads in flight at a time (one trip to the matrix per batch, not
    // per row: the first block of a set often keeps thirty rows)
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int blk = w * 64 + lane;
      const bool take = blk > b && blk < nb;
      unsigned long long km = kept, acc = 0ull;
      while (km) {
        unsigned long long t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          t[u] = 0ull;
          if (km) {
            const int j = __builtin_ctzll(km);
            km &= km - 1ull;
            if (take) t[u] = M[(size_t)(b * 64 + j) * col_blocks + blk];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc |= t[u];
      }
      remv[w] |= acc;
    }
  }
  if (lane == 0) num_keep[set] = kept_total < post_max ? kept_total : post_max;
}

// all-pairs / aligned overlap or IoU (boxes_overlap_kernel, boxes_iou_bev_kernel,
// boxes_aligned_overlap_kernel: iou3d_nms_kernel.cu:236-278)
__global__ __launch_bounds__(256) void pairwise_kernel(const float* __restrict__ a, int na,
                                                       const float* __restrict__ b, int nb,
                                                       int aligned, int want_iou,
                                                       float* __restrict__ out) {
  const long long total = aligned ? na : (long long)na * nb;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int i = aligned ? (int)t : (int)(t / nb);
    const int j = aligned ? (int)t : (int)(t - (long long)i * nb);
    BoxGeom A, Bg;
    make_geom(a + (size_t)i * 7, A);
    make_geom(b + (size_t)j * 7, Bg);
    float v = circles_apart(A, Bg) ? 0.f : box_overlap(A, Bg);
    if (want_iou) v = iou_from_overlap(A, Bg, v);
    out[t] = v;
  }
}

// nms_normal_kernel (iou3d_nms_kernel.cu:327-385): axis-aligned BEV IoU (the heading is ignored), same tiling as
// the rotated mask; every pair is a dozen flops, so no compaction passes
__global__ __launch_bounds__(NMS_THREADS) void nms_normal_mask_kernel(const float* __restrict__ boxes, int n, int col_blocks,
                                                                      float thr, unsigned long long* __restrict__ mask) {
  __shared__ float rowb[NMS_TILE][5], colb[NMS_TILE][5];     // x, y, dx, dy (+1: bank spread)
  __shared__ unsigned long long lmask[NMS_TILE];
  int tile = blockIdx.x, rb = 0, rem = col_blocks;
  while (tile >= rem) { tile -= rem; ++rb; --rem; }
  const int cb = rb + tile, tid = threadIdx.x;
  if (tid < 2 * NMS_TILE) {
    const bool col = tid >= NMS_TILE;
    const int l = tid & (NMS_TILE - 1), g = (col ? cb : rb) * NMS_TILE + l;
    float* d = col ? colb[l] : rowb[l];
    if (g < n) {
      const float* b = boxes + (size_t)g * 7;
      d[0] = b[0]; d[1] = b[1]; d[2] = b[3]; d[3] = b[4];
    }
    if (!col) lmask[l] = 0ull;
  }
  __syncthreads();
  const int r = tid & 63, c0 = (tid >> 6) * 16, gr = rb * NMS_TILE + r;
  unsigned long long bits = 0ull;
  if (gr < n) {
    const float ax = rowb[r][0], ay = rowb[r][1], adx = rowb[r][2], ady = rowb[r][3];
    for (int k = 0; k < 16; ++k) {
      const int c = c0 + k, gc = cb * NMS_TILE + c;
      if (gc < n && gc > gr) {
        const float bx = colb[c][0], by = colb[c][1], bdx = colb[c][2], bdy = colb[c][3];
        const float left = fmaxf(ax - adx / 2, bx - bdx / 2), right = fminf(ax + adx / 2, bx + bdx / 2);
        const float top = fmaxf(ay - ady / 2, by - bdy / 2), bottom = fminf(ay + ady / 2, by + bdy / 2);
        const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
        const float inter = width * height;
        if (inter / fmaxf(adx * ady + bdx * bdy - inter, IOU_EPS) > thr) bits |= 1ull << c;
      }
    }
  }
  if (bits) atomicOr(&lmask[r], bits);
  __syncthreads();
  if (tid < NMS_TILE && rb * NMS_TILE + tid < n) mask[(size_t)(rb * NMS_TILE + tid) * col_blocks + cb] = lmask[tid];
}

static size_t nms_mask_bytes(int num_sets, int cap) {
  const size_t cbk = (size_t)(cap + 63) / 64;
  return pn_align_up((size_t)num_sets * cap * cbk * 8, 256);
}
// workspace: bit matrix | box geometry | queue counter | clip queue
extern "C" size_t pn_nms_workspace_bytes(int num_sets, int cap) {
  if (num_sets <= 0 || cap <= 0) return 0;
  return nms_mask_bytes(num_sets, cap) + pn_align_up((size_t)num_sets * cap * sizeof(BoxGeom), 256) + 256 +
         (size_t)NMS_QUEUE_CAP * sizeof(NmsPair);
}

// mask-only workspaces (with_queue == false: the static entry point) clip inside the tiles
static int nms_run(const float* boxes, int num_sets, int cap, const int32_t* counts_dev,
                   int count_host, const float* thresh_dev, float thresh,
                   const int32_t* post_max_dev, int post_max, int keep_stride,
                   int32_t* keep, int32_t* num_keep, void* workspace,
                   size_t workspace_bytes, bool with_queue, hipStream_t stream) {
  PN_CHECK_ARG(num_sets >= 1 && cap >= 1 && cap <= PN_NMS_MAX_BOXES, "pn_nms_rotated: sets %d cap %d (cap <= %d)",
               num_sets, cap, PN_NMS_MAX_BOXES);
  PN_CHECK_ARG(boxes && keep && num_keep && workspace, "pn_nms_rotated: null argument");
  PN_CHECK_ARG(counts_dev || (count_host >= 1 && count_host <= cap),
               "pn_nms_rotated: need device counts or a host count in 1..cap (got %d)", count_host);
  PN_CHECK_ARG(keep_stride >= 1 && (post_max_dev || post_max >= 1), "pn_nms_rotated: bad post_max");
  const size_t need = with_queue ? pn_nms_workspace_bytes(num_sets, cap) : nms_mask_bytes(num_sets, cap);
  if (workspace_bytes < need) {
    pn_set_error("pn_nms_rotated: workspace %zu < %zu bytes", workspace_bytes, need);
    return PN_ENOSPC;
  }
  BoxGeom* geoms = nullptr;
  int* qcount = nullptr;
  NmsPair* queue = nullptr;
  if (with_queue) {
    char* w = (char*)workspace + nms_mask_bytes(num_sets, cap);
    geoms = (BoxGeom*)w;
    w += pn_align_up((size_t)num_sets * cap * sizeof(BoxGeom), 256);
    qcount = (int*)w;
    queue = (NmsPair*)(w + 256);
    const int rc_ = pn_fill_async(qcount, 0, 256, stream);
    if (rc_ != PN_OK) return rc_;
  }
  const int n_for_grid = counts_dev ? cap : count_host;
  const int cbk_store = (cap + 63) / 64;
  const int cbk_grid = (n_for_grid + 63) / 64;
  // tile unranking walks the triangle of the *stored* width, so launch that triangle's tiles
  // whose rows/cols can be populated; tiles past n return immediately
  (void)cbk_grid;
  const int tiles = cbk_store * (cbk_store + 1) / 2;
  unsigned long long* mask = (unsigned long long*)workspace;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(tiles * num_sets), dim3(NMS_THREADS), 0, stream, boxes,
                     counts_dev, counts_dev ? 0 : count_host, cap, cbk_store, tiles, thresh_dev,
                     thresh, mask, geoms, qcount, queue, NMS_QUEUE_CAP);
  PN_CHECK_LAUNCH("nms_mask_kernel");
  if (with_queue) {
    hipLaunchKernelGGL(nms_clip_kernel, dim3(2048), dim3(64), 0, stream, geoms, queue, qcount, NMS_QUEUE_CAP, cap,
                       cbk_store, thresh_dev, thresh, mask);
    PN_CHECK_LAUNCH("nms_clip_kernel");
  }
  if (cap <= 4096)
    hipLaunchKernelGGL(nms_select_kernel<1>, dim3(num_sets), dim3(64), 0, stream, mask, counts_dev,
                       counts_dev ? 0 : count_host, cap, cbk_store, post_max_dev, post_max,
                       keep_stride, keep, num_keep);
  else
    hipLaunchKernelGGL((nms_select_kernel<PN_NMS_MAX_BOXES / 4096>), dim3(num_sets), dim3(64), 0, stream, mask,
                       counts_dev, counts_dev ? 0 : count_host, cap, cbk_store, post_max_dev, post_max,
                       keep_stride, keep, num_keep);
  PN_CHECK_LAUNCH("nms_select_kernel");
  return PN_OK;
}

extern "C" int pn_nms_rotated(const float* boxes, int num_sets, int cap, const int32_t* counts_dev,
                              int count_host, const float* thresh_dev, float thresh,
                              const int32_t* post_max_dev, int post_max, int keep_stride,
                              int32_t* keep, int32_t* num_keep, void* workspace,
                              size_t workspace_bytes, pnStream stream_) {
  return nms_run(boxes, num_sets, cap, counts_dev, count_host, thresh_dev, thresh, post_max_dev, post_max,
                 keep_stride, keep, num_keep, workspace, workspace_bytes, true, (hipStream_t)stream_);
}

extern "C" int pn_nms_normal(const float* boxes, int n, float thresh, int post_max, int32_t* keep, int32_t* num_keep,
                             void* workspace, size_t workspace_bytes, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(boxes && keep && num_keep && workspace && n >= 1 && n <= PN_NMS_MAX_BOXES && post_max >= 1,
               "pn_nms_normal: bad argument (n %d, at most %d)", n, PN_NMS_MAX_BOXES);
  if (workspace_bytes < nms_mask_bytes(1, n)) {
    pn_set_error("pn_nms_normal: workspace %zu < %zu bytes", workspace_bytes, nms_mask_bytes(1, n));
    return PN_ENOSPC;
  }
  const int cbk = (n + 63) / 64;
  unsigned long long* mask = (unsigned long long*)workspace;
  hipLaunchKernelGGL(nms_normal_mask_kernel, dim3(cbk * (cbk + 1) / 2), dim3(NMS_THREADS), 0, stream, boxes, n, cbk, thresh,
                     mask);
  PN_CHECK_LAUNCH("nms_normal_mask_kernel");
  if (n <= 4096)
    hipLaunchKernelGGL(nms_select_kernel<1>, dim3(1), dim3(64), 0, stream, mask, nullptr, n, n, cbk, nullptr, post_max, post_max,
                       keep, num_keep);
  else
    hipLaunchKernelGGL((nms_select_kernel<PN_NMS_MAX_BOXES / 4096>), dim3(1), dim3(64), 0, stream, mask, nullptr, n, n, cbk,
                       nullptr, post_max, post_max, keep, num_keep);
  PN_CHECK_LAUNCH("nms_select_kernel");
  return PN_OK;
}

// ------------------------------------------------------------------------------------------------
// pn_nms_rotated for callers that may not allocate: the suppression bit matrix lives in a
// file-scope device array instead of a caller workspace.  Each GPU holds its own copy of the array
// at its own address; the address is resolved once per device and cached.  The bit matrix needs no
// clearing between calls: nms_mask_kernel writes every word nms_select_kernel reads.
static constexpr size_t NMS_STATIC_BYTES = PN_NMS_STATIC_BYTES;   // 16 sets x 1024 boxes, or 1 x 4096
static constexpr int NMS_STATIC_MAX_DEVICES = 64;
static __device__ __align__(256) unsigned char scratch[NMS_STATIC_BYTES];
static void* g_static_mask[NMS_STATIC_MAX_DEVICES];
static std::mutex g_static_mask_lock;

static int nms_static_mask(void** out) {
  int dev = -1;
  PN_CHECK_HIP(hipGetDevice(&dev));
  PN_CHECK_ARG(dev >= 0 && dev < NMS_STATIC_MAX_DEVICES, "pn_nms_rotated_static: device %d out of range", dev);
  std::lock_guard<std::mutex> guard(g_static_mask_lock);
  if (!g_static_mask[dev]) {
    void* p = nullptr;
    PN_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    PN_CHECK_ARG(p != nullptr, "pn_nms_rotated_static: no device address for the static bit matrix");
    g_static_mask[dev] = p;
  }
  *out = g_static_mask[dev];
  return PN_OK;
}

extern "C" int pn_nms_rotated_static(const float* boxes, int num_sets, int cap,
                                     const int32_t* counts_dev, int count_host,
                                     const float* thresh_dev, float thresh,
                                     const int32_t* post_max_dev, int post_max, int keep_stride,
                                     int32_t* keep, int32_t* num_keep, pnStream stream_) {
  PN_CHECK_ARG(num_sets >= 1 && cap >= 1 && cap <= 4096,
               "pn_nms_rotated_static: sets %d cap %d (cap <= 4096)", num_sets, cap);
  const size_t need = nms_mask_bytes(num_sets, cap);
  if (need > NMS_STATIC_BYTES) {
    pn_set_error("pn_nms_rotated_static: %d sets x %d boxes need a %zu-byte bit matrix, the static "
                 "one holds %zu bytes; use pn_nms_rotated with a caller workspace",
                 num_sets, cap, need, NMS_STATIC_BYTES);
    return PN_ENOSPC;
  }
  void* mask = nullptr;
  const int rc = nms_static_mask(&mask);
  if (rc != PN_OK) return rc;
  return nms_run(boxes, num_sets, cap, counts_dev, count_host, thresh_dev, thresh, post_max_dev, post_max,
                 keep_stride, keep, num_keep, mask, NMS_STATIC_BYTES, false, (hipStream_t)stream_);
}

extern "C" int pn_boxes_pairwise_bev(const float* boxes_a, int na, const float* boxes_b, int nb,
                                     int aligned, int want_iou, float* out, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(na >= 0 && nb >= 0 && out, "pn_boxes_pairwise_bev: bad argument");
  PN_CHECK_ARG(!aligned || na == nb, "pn_boxes_pairwise_bev: aligned needs na == nb");
  const long long total = aligned ? na : (long long)na * nb;
  if (total == 0) return PN_OK;
  PN_CHECK_ARG(boxes_a && boxes_b, "pn_boxes_pairwise_bev: null boxes");
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pairwise_kernel, dim3((int)blocks), dim3(256), 0, stream, boxes_a, na, boxes_b,
                     nb, aligned, want_iou, out);
  PN_CHECK_LAUNCH("pairwise_kernel");
  return PN_OK;
}
