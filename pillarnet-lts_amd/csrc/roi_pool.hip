// "RoI pillar pooling" of Pillar R-CNN: a G x G grid of points laid over every rotated RoI, each
// point bilinearly interpolated from a dense BEV feature map.
//
// Replaces, in one launch (the reference runs ~30 small torch kernels and builds four [N, C]
// gathers per sample):
//   det3d/core/bbox/box_torch_ops.py:220-251   center_to_grid_box2d / get_dense_roi_grid_points
//   det3d/core/bbox/box_torch_ops.py:159-172   rotation_2d
//   det3d/core/utils/center_utils.py:91-120    bilinear_interpolate_torch
//   det3d/models/second_stage/bev_interpolation.py:236-268  get_pooling_points / interpolate_from_bev_features
//
// One wavefront per (roi, grid point): the 4 corner rows of the channels-last map are contiguous C
// vectors, so lanes stride over channels with coalesced reads.
#include "common.h"

// rois [B, R, roi_dim] (x, y, z, dx, dy, dz, ..., heading last), map [B, H*W, map_ld] channels-last rows
template <typename T>
__global__ __launch_bounds__(256) void roi_grid_pool_batched_kernel(
    const float* __restrict__ rois, int roi_dim, int B, int R, int G, const T* __restrict__ map, int map_ld, int H,
    int W, int C, float x_min, float y_min, float inv_cell, float* __restrict__ out, float* __restrict__ points) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long long total = (long long)B * R * G * G;
  if (wave >= total) return;
  const int gp = (int)(wave % (G * G));
  const long long r = wave / (G * G);        // b*R + roi
  const int b = (int)(r / R);
  const float* roi = rois + r * roi_dim;
  const float cx = roi[0], cy = roi[1], dx = roi[3], dy = roi[4], ang = roi[roi_dim - 1];
  const int i = gp / G, j = gp % G;
  const float lx = ((float)i + 0.5f) / (float)G * dx - dx / 2;
  const float ly = ((float)j + 0.5f) / (float)G * dy - dy / 2;
  const float s = sinf(ang), c = cosf(ang);
  const float px = lx * c + ly * s + cx;
  const float py = -lx * s + ly * c + cy;
  if (points && lane == 0) {
    points[wave * 2 + 0] = px;
    points[wave * 2 + 1] = py;
  }
  const float x = (px - x_min) * inv_cell, y = (py - y_min) * inv_cell;
  int x0 = (int)floorf(x), y0 = (int)floorf(y);
  int x1 = x0 + 1, y1 = y0 + 1;
  x0 = min(max(x0, 0), W - 1); x1 = min(max(x1, 0), W - 1);
  y0 = min(max(y0, 0), H - 1); y1 = min(max(y1, 0), H - 1);
  // weights from the CLAMPED corners, as bilinear_interpolate_torch computes them
  const float wa = ((float)x1 - x) * ((float)y1 - y), wb = ((float)x1 - x) * (y - (float)y0);
  const float wc = (x - (float)x0) * ((float)y1 - y), wd = (x - (float)x0) * (y - (float)y0);
  const T* base = map + (size_t)b * H * W * map_ld;
  const T* Ia = base + ((size_t)y0 * W + x0) * map_ld;
  const T* Ib = base + ((size_t)y1 * W + x0) * map_ld;
  const T* Ic = base + ((size_t)y0 * W + x1) * map_ld;
  const T* Id = base + ((size_t)y1 * W + x1) * map_ld;
  float* o = out + wave * C;
  for (int ch = lane; ch < C; ch += 64)
    o[ch] = (float)Ia[ch] * wa + (float)Ib[ch] * wb + (float)Ic[ch] * wc + (float)Id[ch] * wd;
}

extern "C" int pn_roi_grid_pool(int dtype, const float* rois, int roi_dim, int batch, int rois_per_sample,
                                int grid, const void* map, int map_ld, int H, int W, int C, float x_min,
                                float y_min, float cell_size, float* out, float* points,
                                pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(dtype == PN_DTYPE_F32 || dtype == PN_DTYPE_BF16, "pn_roi_grid_pool: dtype %d", dtype);
  PN_CHECK_ARG(rois && map && out && roi_dim >= 7 && batch >= 1 && rois_per_sample >= 0 && grid >= 1 && C >= 1 &&
                   map_ld >= C && H >= 1 && W >= 1 && cell_size > 0.f,
               "pn_roi_grid_pool: bad argument");
  const long long waves = (long long)batch * rois_per_sample * grid * grid;
  if (waves == 0) return PN_OK;
  const int blocks = (int)((waves + 3) / 4);
  const float inv = 1.0f / cell_size;
  if (dtype == PN_DTYPE_BF16)
    hipLaunchKernelGGL(roi_grid_pool_batched_kernel<__bf16>, dim3(blocks), dim3(256), 0, stream, rois, roi_dim, batch,
                       rois_per_sample, grid, (const __bf16*)map, map_ld, H, W, C, x_min, y_min, inv, out, points);
  else
    hipLaunchKernelGGL(roi_grid_pool_batched_kernel<float>, dim3(blocks), dim3(256), 0, stream, rois, roi_dim, batch,
                       rois_per_sample, grid, (const float*)map, map_ld, H, W, C, x_min, y_min, inv, out, points);
  PN_CHECK_LAUNCH("roi_grid_pool_batched_kernel");
  return PN_OK;
}

// Gradient of the pooling w.r.t. the map: every grid point adds grad_out * (its four bilinear weights) to its four
// corner rows of grad_map (fp32 atomics; the reference's autograd does the same through index_put/gather
// backward of bilinear_interpolate_torch, center_utils.py:91-120).  grad_map must be zero on entry.
__global__ __launch_bounds__(256) void roi_grid_pool_grad_kernel(
    const float* __restrict__ rois, int roi_dim, int B, int R, int G, const float* __restrict__ grad_out, int H,
    int W, int C, float x_min, float y_min, float inv_cell, float* __restrict__ grad_map) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long long total = (long long)B * R * G * G;
  if (wave >= total) return;
  const int gp = (int)(wave % (G * G));
  const long long r = wave / (G * G);
  const int b = (int)(r / R);
  const float* roi = rois + r * roi_dim;
  const float cx = roi[0], cy = roi[1], dx = roi[3], dy = roi[4], ang = roi[roi_dim - 1];
  const int i = gp / G, j = gp % G;
  const float lx = ((float)i + 0.5f) / (float)G * dx - dx / 2;
  const float ly = ((float)j + 0.5f) / (float)G * dy - dy / 2;
  const float s = sinf(ang), c = cosf(ang);
  const float px = lx * c + ly * s + cx;
  const float py = -lx * s + ly * c + cy;
  const float x = (px - x_min) * inv_cell, y = (py - y_min) * inv_cell;
  int x0 = (int)floorf(x), y0 = (int)floorf(y);
  int x1 = x0 + 1, y1 = y0 + 1;
  x0 = min(max(x0, 0), W - 1); x1 = min(max(x1, 0), W - 1);
  y0 = min(max(y0, 0), H - 1); y1 = min(max(y1, 0), H - 1);
  const float wa = ((float)x1 - x) * ((float)y1 - y), wb = ((float)x1 - x) * (y - (float)y0);
  const float wc = (x - (float)x0) * ((float)y1 - y), wd = (x - (float)x0) * (y - (float)y0);
  float* base = grad_map + (size_t)b * H * W * C;
  float* Ga = base + ((size_t)y0 * W + x0) * C;
  float* Gb = base + ((size_t)y1 * W + x0) * C;
  float* Gc = base + ((size_t)y0 * W + x1) * C;
  float* Gd = base + ((size_t)y1 * W + x1) * C;
  const float* g = grad_out + wave * C;
  for (int ch = lane; ch < C; ch += 64) {
    const float v = g[ch];
    atomicAdd(Ga + ch, v * wa);
    atomicAdd(Gb + ch, v * wb);
    atomicAdd(Gc + ch, v * wc);
    atomicAdd(Gd + ch, v * wd);
  }
}

extern "C" int pn_roi_grid_pool_grad(const float* rois, int roi_dim, int batch, int rois_per_sample, int grid,
                                     const float* grad_out, int H, int W, int C, float x_min, float y_min,
                                     float cell_size, float* grad_map, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PN_CHECK_ARG(rois && grad_out && grad_map && roi_dim >= 7 && batch >= 1 && rois_per_sample >= 0 && grid >= 1 &&
                   C >= 1 && H >= 1 && W >= 1 && cell_size > 0.f,
               "pn_roi_grid_pool_grad: bad argument");
  const long long waves = (long long)batch * rois_per_sample * grid * grid;
  if (waves == 0) return PN_OK;
  hipLaunchKernelGGL(roi_grid_pool_grad_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, rois, roi_dim,
                     batch, rois_per_sample, grid, grad_out, H, W, C, x_min, y_min, 1.0f / cell_size, grad_map);
  PN_CHECK_LAUNCH("roi_grid_pool_grad_kernel");
  return PN_OK;
}

// ---- points_in_boxes_2d (det3d/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-21,39-50,354-377) ----
// boxes [B, T, 7] in pcdet convention (x, y, z, dx, dy, dz, heading), pts [B, M, 2]; out [B, M] = index of the
// FIRST box whose BEV rectangle (+1e-5 margin) holds the point, else -1.
__global__ __launch_bounds__(256) void points_in_boxes_2d_kernel(int boxes_num, int pts_num,
                                                                 const float* __restrict__ boxes,
                                                                 const float* __restrict__ pts,
                                                                 int32_t* __restrict__ out) {
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= pts_num) return;
  const float* bx = boxes + (size_t)b * boxes_num * 7;
  const float x = pts[((size_t)b * pts_num + p) * 2], y = pts[((size_t)b * pts_num + p) * 2 + 1];
  int hit = -1;
  for (int k = 0; k < boxes_num; ++k) {
    const float* q = bx + k * 7;
    const float sx = x - q[0], sy = y - q[1];
    const float cosa = cosf(-q[6]), sina = sinf(-q[6]);
    const float lx = sx * cosa + sy * (-sina), ly = sx * sina + sy * cosa;
    // the CUDA source compares against double expressions (dx / 2.0 + MARGIN with float MARGIN)
    if (fabs((double)lx) < (double)q[3] / 2.0 + (double)1e-5f && fabs((double)ly) < (double)q[4] / 2.0 + (double)1e-5f) {
      hit = k;
      break;
    }
  }
  out[(size_t)b * pts_num + p] = hit;
}

extern "C" int pn_points_in_boxes_2d(const float* boxes, const float* pts, int batch, int boxes_num, int pts_num,
                                     int32_t* box_idx_of_points, pnStream stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (batch == 0 || pts_num == 0) return PN_OK;
  PN_CHECK_ARG(pts && box_idx_of_points && batch > 0 && boxes_num >= 0 && (boxes || boxes_num == 0),
               "pn_points_in_boxes_2d: null argument");
  hipLaunchKernelGGL(points_in_boxes_2d_kernel, dim3(pn_div_up(pts_num, 256), batch), dim3(256), 0, stream, boxes_num,
                     pts_num, boxes, pts, box_idx_of_points);
  PN_CHECK_LAUNCH("points_in_boxes_2d_kernel");
  return PN_OK;
}
