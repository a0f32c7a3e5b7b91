// Host-side hard voxelizer: the `Voxelization` pipeline step of the legacy configs
// (det3d/datasets/pipelines `Voxelization(cfg=voxel_generator)` ->
// det3d/ops/point_cloud/point_cloud_ops.py:112-184 points_to_voxel, numba kernels :8-109).
// The reference runs this on the CPU inside its dataloader workers; so does this routine (no device
// work: it is sequential by definition -- voxels are numbered in the order their first point arrives).
#include "common.h"
#include <math.h>
#include <string.h>

// points [n, feats] fp32; voxel_size[3], range[6] fp32; grid_map: int32 per voxel of the grid
// (z,y,x order when reverse_index), all -1 on entry; voxels [max_voxels, max_points, feats] zeroed,
// coors [max_voxels, 3], num_points [max_voxels] zeroed.  Returns the number of voxels, or PN_EINVAL.
extern "C" int pn_points_to_voxel(const float* points, long long n, int feats, const float* voxel_size,
                                  const float* range, int reverse_index, int max_points, int max_voxels,
                                  int32_t* grid_map, float* voxels, int32_t* coors, int32_t* num_points) {
  PN_CHECK_ARG(points || n == 0, "pn_points_to_voxel: null points");
  PN_CHECK_ARG(voxel_size && range && grid_map && voxels && coors && num_points && feats >= 3 && max_points >= 1 &&
                   max_voxels >= 1 && n >= 0,
               "pn_points_to_voxel: bad argument");
  int grid[3], dims[3];
  for (int a = 0; a < 3; ++a) {
    grid[a] = (int)rintf((range[3 + a] - range[a]) / voxel_size[a]);
    PN_CHECK_ARG(grid[a] >= 1, "pn_points_to_voxel: empty grid along axis %d", a);
  }
  for (int a = 0; a < 3; ++a) dims[a] = reverse_index ? grid[2 - a] : grid[a];
  int count = 0;
  for (long long i = 0; i < n; ++i) {
    const float* p = points + i * feats;
    int c[3];
    bool inside = true;
    for (int a = 0; a < 3 && inside; ++a) {
      const float v = floorf((p[a] - range[a]) / voxel_size[a]);
      inside = v >= 0.f && v < (float)grid[a];
      if (inside) c[reverse_index ? 2 - a : a] = (int)v;
    }
    if (!inside) continue;
    int32_t& slot = grid_map[((long long)c[0] * dims[1] + c[1]) * dims[2] + c[2]];
    if (slot < 0) {
      if (count >= max_voxels) continue;   // the voxel budget is spent: the point is dropped
      slot = count;
      memcpy(coors + 3 * (long long)count, c, sizeof(c));
      ++count;
    }
    int32_t& filled = num_points[slot];
    if (filled < max_points) {
      memcpy(voxels + ((long long)slot * max_points + filled) * feats, p, sizeof(float) * feats);
      ++filled;
    }
  }
  return count;
}
