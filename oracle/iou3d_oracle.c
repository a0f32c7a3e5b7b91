/*
 * CPU restatement (plain C, fp32) of the reference's rotated-BEV IoU and greedy NMS.
 * TEST INFRASTRUCTURE ONLY -- used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg as the checker; the product never links or calls it.
 *
 * Pinned against the reference itself: oracle/_ref/ is the reference's own
 * det3d/ops/iou3d_nms/src/iou3d_cpu.cpp compiled in place (oracle/build_ref.py);
 * tests/test_oracle_cpu.py checks this file against it bit for bit, and tests/golden/iou3d_*.npz
 * holds its outputs for the GPU box, where /root/reference does not exist.
 *
 * Follows:
 *   det3d/ops/iou3d_nms/src/iou3d_cpu.cpp:59-229            cross / intersection / box_overlap / iou_bev
 *   det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:36-234      (device twin, same arithmetic)
 *   det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:280-324     nms_kernel: bit (i, j) set iff j > i and IoU > thr
 *   det3d/ops/iou3d_nms/src/iou3d_nms.cpp:139-156           greedy pass over the bit matrix
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (no FMA contraction: every product and sum is
 * rounded separately, as in the reference's x86 build).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IOU_EPS 1e-8f
#define IN_BOX_MARGIN 1e-2f
#define MAX_POLY 24 /* 16 edge crossings + 8 corners; the reference reserves 16 */

typedef struct { float x, y; } pt2;

static float cross3(pt2 p1, pt2 p2, pt2 p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

static float cross2(pt2 a, pt2 b) { return a.x * b.y - a.y * b.x; }

static float fmin2(float a, float b) { return a > b ? b : a; }
static float fmax2(float a, float b) { return a > b ? a : b; }

/* axis-aligned bounding boxes of the two segments overlap (iou3d_cpu.cpp:67-73) */
static int seg_bbox_overlap(pt2 p1, pt2 p2, pt2 q1, pt2 q2) {
  return fmin2(p1.x, p2.x) <= fmax2(q1.x, q2.x) && fmin2(q1.x, q2.x) <= fmax2(p1.x, p2.x) &&
         fmin2(p1.y, p2.y) <= fmax2(q1.y, q2.y) && fmin2(q1.y, q2.y) <= fmax2(p1.y, p2.y);
}

/* proper crossing of segment p0p1 with q0q1 (iou3d_cpu.cpp:87-114) */
static int seg_intersection(pt2 p1, pt2 p0, pt2 q1, pt2 q0, pt2* out) {
  if (!seg_bbox_overlap(p0, p1, q0, q1)) return 0;
  float s1 = cross3(q0, p1, p0);
  float s2 = cross3(p1, q1, p0);
  float s3 = cross3(p0, q1, q0);
  float s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > IOU_EPS) {
    out->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    out->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    out->x = (b0 * c1 - b1 * c0) / D;
    out->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

/* point inside the rotated rectangle, 1 cm slack (iou3d_cpu.cpp:75-85) */
static int point_in_box(const float* box, pt2 p) {
  float cx = box[0], cy = box[1];
  float c = cosf(-box[6]), s = sinf(-box[6]);
  float rx = (p.x - cx) * c + (p.y - cy) * (-s);
  float ry = (p.x - cx) * s + (p.y - cy) * c;
  return fabsf(rx) < box[3] / 2 + IN_BOX_MARGIN && fabsf(ry) < box[4] / 2 + IN_BOX_MARGIN;
}

static pt2 rotate_about(pt2 ctr, float c, float s, pt2 p) {
  pt2 r;
  r.x = (p.x - ctr.x) * c + (p.y - ctr.y) * (-s) + ctr.x;
  r.y = (p.x - ctr.x) * s + (p.y - ctr.y) * c + ctr.y;
  return r;
}

static void box_corners(const float* box, pt2 out[5]) {
  float hx = box[3] / 2, hy = box[4] / 2;
  float x1 = box[0] - hx, y1 = box[1] - hy, x2 = box[0] + hx, y2 = box[1] + hy;
  pt2 ctr = {box[0], box[1]};
  float c = cosf(box[6]), s = sinf(box[6]);
  pt2 raw[4] = {{x1, y1}, {x2, y1}, {x2, y2}, {x1, y2}};
  for (int k = 0; k < 4; ++k) out[k] = rotate_about(ctr, c, s, raw[k]);
  out[4] = out[0];
}

/* area of the intersection polygon of two rotated rectangles (iou3d_cpu.cpp:128-220) */
float pn_oracle_box_overlap(const float* a, const float* b) {
  pt2 ca[5], cb[5], poly[MAX_POLY], ctr = {0.f, 0.f};
  int cnt = 0;
  box_corners(a, ca);
  box_corners(b, cb);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (cnt < MAX_POLY && seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], &poly[cnt])) {
        ctr.x = ctr.x + poly[cnt].x;
        ctr.y = ctr.y + poly[cnt].y;
        ++cnt;
      }
  for (int k = 0; k < 4; ++k) {
    if (cnt < MAX_POLY && point_in_box(a, cb[k])) {
      ctr.x = ctr.x + cb[k].x; ctr.y = ctr.y + cb[k].y;
      poly[cnt++] = cb[k];
    }
    if (cnt < MAX_POLY && point_in_box(b, ca[k])) {
      ctr.x = ctr.x + ca[k].x; ctr.y = ctr.y + ca[k].y;
      poly[cnt++] = ca[k];
    }
  }
  ctr.x /= cnt;
  ctr.y /= cnt;
  /* bubble sort by polar angle about the centroid (iou3d_cpu.cpp:116-118,199-209); the angle of
     a vertex does not change while it moves, so it is evaluated once per vertex */
  float ang[MAX_POLY];
  for (int k = 0; k < cnt; ++k) ang[k] = atan2f(poly[k].y - ctr.y, poly[k].x - ctr.x);
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (ang[i] > ang[i + 1]) {
        pt2 t = poly[i]; poly[i] = poly[i + 1]; poly[i + 1] = t;
        float u = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = u;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    pt2 u = {poly[k].x - poly[0].x, poly[k].y - poly[0].y};
    pt2 v = {poly[k + 1].x - poly[0].x, poly[k + 1].y - poly[0].y};
    area += cross2(u, v);
  }
  return fabsf(area) / 2.0f;
}

float pn_oracle_iou_bev(const float* a, const float* b) {
  float sa = a[3] * a[4], sb = b[3] * b[4];
  float ov = pn_oracle_box_overlap(a, b);
  return ov / fmaxf(sa + sb - ov, IOU_EPS);
}

/* boxes_iou_bev_cpu (iou3d_cpu.cpp:232-254) / boxes_overlap: out[i*nb + j] */
void pn_oracle_iou_matrix(const float* a, int na, const float* b, int nb, float* out, int overlap_only) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j)
      out[(size_t)i * nb + j] = overlap_only ? pn_oracle_box_overlap(a + i * 7, b + j * 7)
                                             : pn_oracle_iou_bev(a + i * 7, b + j * 7);
}

/* nms_gpu (iou3d_nms.cpp:113-159): boxes are already sorted by score; row i suppresses j > i when
   IoU(i, j) > thresh; rows are visited in order.  keep[] receives the kept positions, the return
   value is their number.  min_gap (may be NULL) receives min |IoU - thresh| over the pairs the
   greedy pass actually consulted -- the tests use it to assert the keep set is not borderline. */
int pn_oracle_nms(const float* boxes, int n, float thresh, int64_t* keep, float* min_gap) {
  unsigned char* removed = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
  int kept = 0;
  float gap = INFINITY;
  for (int i = 0; i < n; ++i) {
    if (removed[i]) continue;
    keep[kept++] = i;
    for (int j = i + 1; j < n; ++j) {
      float iou = pn_oracle_iou_bev(boxes + i * 7, boxes + j * 7);
      float d = fabsf(iou - thresh);
      if (!removed[j] && d < gap) gap = d;
      if (iou > thresh) removed[j] = 1;
    }
  }
  free(removed);
  if (min_gap) *min_gap = gap;
  return kept;
}
