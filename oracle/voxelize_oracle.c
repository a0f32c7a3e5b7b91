/*
 * CPU restatement of the reference's hard voxelizer ("det3d CPU voxelizer"):
 *   det3d/ops/point_cloud/point_cloud_ops.py:8-55    _points_to_voxel_reverse_kernel (zyx coords)
 *   det3d/ops/point_cloud/point_cloud_ops.py:59-109   _points_to_voxel_kernel (xyz coords)
 *   det3d/ops/point_cloud/point_cloud_ops.py:112-184  points_to_voxel (buffers, truncation)
 *
 * TEST INFRASTRUCTURE ONLY (the checker and the CPU-baseline leg of the benchmark); the product
 * never links or loads it.  Semantics kept exactly: first-come voxel order, a point whose voxel
 * would be number max_voxels or later is dropped, at most max_points points are stored per voxel
 * (later ones dropped), all arithmetic in the points' dtype (fp32): c = floor((p - lo) / size).
 * Pinned by tests/golden/voxelize.npz = outputs of the reference's own points_to_voxel
 * (numba's jit replaced by a pass-through, so the same Python ran interpreted).
 * Build with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* points [n, ndim] fp32; voxel_size[3], range[6] fp32; coor_to_voxelidx: int32 map of the voxel
 * grid, every entry -1 on entry (shape grid[2],grid[1],grid[0] when reverse_index else
 * grid[0],grid[1],grid[2]); voxels [max_voxels, max_points, ndim] zeroed; coors [max_voxels, 3];
 * num_points_per_voxel [max_voxels] zeroed.  Returns the number of voxels. */
int pn_oracle_points_to_voxel(const float* points, int64_t n, int ndim_feat,
                              const float* voxel_size, const float* range, int reverse_index,
                              int max_points, int max_voxels, int32_t* coor_to_voxelidx,
                              float* voxels, int32_t* coors, int32_t* num_points_per_voxel) {
  int32_t grid[3];
  for (int j = 0; j < 3; ++j) {
    const float g = (range[3 + j] - range[j]) / voxel_size[j];
    grid[j] = (int32_t)rintf(g); /* np.round: half to even, as rintf in the default mode */
  }
  int32_t shape[3];
  for (int j = 0; j < 3; ++j) shape[j] = reverse_index ? grid[2 - j] : grid[j];
  int voxel_num = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float* p = points + i * ndim_feat;
    int32_t coor[3];
    int failed = 0;
    for (int j = 0; j < 3; ++j) {
      const float c = floorf((p[j] - range[j]) / voxel_size[j]);
      if (c < 0 || c >= (float)grid[j]) {
        failed = 1;
        break;
      }
      coor[reverse_index ? 2 - j : j] = (int32_t)c;
    }
    if (failed) continue;
    const int64_t cell = ((int64_t)coor[0] * shape[1] + coor[1]) * shape[2] + coor[2];
    int32_t v = coor_to_voxelidx[cell];
    if (v == -1) {
      v = voxel_num;
      if (voxel_num >= max_voxels) continue;
      ++voxel_num;
      coor_to_voxelidx[cell] = v;
      coors[3 * v + 0] = coor[0];
      coors[3 * v + 1] = coor[1];
      coors[3 * v + 2] = coor[2];
    }
    const int32_t num = num_points_per_voxel[v];
    if (num < max_points) {
      memcpy(voxels + ((int64_t)v * max_points + num) * ndim_feat, p, sizeof(float) * ndim_feat);
      num_points_per_voxel[v] = num + 1;
    }
  }
  return voxel_num;
}
