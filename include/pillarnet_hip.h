/*
 * pillarnet_hip.h -- C ABI of the MI355X (gfx950) pillar-detection hot path.
 *
 * Every entry point takes plain device pointers, sizes and a HIP stream
 * (hipStream_t passed as void*).  Nothing here allocates, frees or
 * synchronises: workspaces are caller-owned, counts that are only known on
 * the device (number of non-empty pillars, number of kept boxes ...) stay on
 * the device in caller-provided int32 slots, so a whole frame can be queued
 * (or captured into a hipGraph) without a host round trip.
 *
 * Return value: 0 on success, otherwise a negative PN_E* code or a positive
 * hipError_t; pn_last_error() returns a thread-local description.  (The
 * reference's launchers call exit(-1) on a launch error instead:
 * det3d/ops/pillar_ops/src/pillar_ops_gpu.cu:53-57.)
 *
 * Each function cites the reference interface it replaces (paths relative to
 * the reference root).
 */
#ifndef PILLARNET_HIP_H_
#define PILLARNET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PN_OK 0
#define PN_EINVAL (-1)   /* bad argument (shape / alignment / capacity)   */
#define PN_ENOSPC (-2)   /* workspace too small                            */
#define PN_MAX_BATCH 32  /* batch offsets travel as a kernel argument      */
#define PN_DTYPE_F32 0   /* feature / weight element type of the sparse ops */
#define PN_DTYPE_BF16 1

typedef void* pnStream;  /* hipStream_t */

int pn_abi_version(void);
const char* pn_last_error(void);

/* ------------------------------------------------------------------------
 * (a1-a4) point cloud -> pillars, fused.
 *
 * Replaces, in one call:
 *   DynamicPFE.forward                det3d/models/readers/dynamic_pillar_encoder.py:29-50
 *   PillarQueryAndGroup.forward       det3d/ops/pillar_ops/pillar_utils.py:22-58
 *     pillar_cuda.create_point_pillar_index_stack_wrapper   src/pillar_ops_gpu.cu:13-58
 *     pillar_cuda.create_pillar_indices_wrapper              src/pillar_ops_gpu.cu:60-94
 *     pillar_cuda.gather_indice_wrapper / gather_feature_wrapper  src/group_ops_gpu.cu:8-78
 *   PillarMaxPooling.forward (eval)   det3d/ops/pillar_ops/pillar_modules.py:56-74
 *     Linear(F+2 -> C, no bias) + BatchNorm1d(eval) + ReLU, then
 *     pillar_cuda.scatter_max_wrapper                        src/scatter_ops_gpu.cu:13-60
 * ------------------------------------------------------------------------ */

typedef struct {
  int32_t batch;           /* B, 1..PN_MAX_BATCH                              */
  int32_t height, width;   /* BEV grid (pillar_utils.py:7-10)                 */
  int32_t point_feats;     /* F: floats per point row (5 nuScenes, 5 Waymo)   */
  int32_t out_channels;    /* C: PFN width, must be 32                        */
  float x_min, y_min;      /* pc_range[0], pc_range[1] as fp32                */
  float inv_pillar;        /* fp32(1) / fp32(pillar_size): torch's device div-by-scalar path */
  float pillar_size;       /* fp32(pillar_size)                               */
  float x_offset, y_offset;/* fp32(pillar_size/2 + pc_range[0|1]), computed in double */
  int32_t true_divide;     /* 1: floor((x-min)/ps) (torch-CPU semantics) instead of *inv */
} pnPillarGrid;

/* bytes of scratch pn_pillarize needs for `n_points` points on this grid. */
size_t pn_pillarize_workspace_bytes(const pnPillarGrid* g, int64_t n_points);

/*
 * points        [n_points, F] fp32, samples concatenated; sample b owns rows
 *               [batch_offsets[b], batch_offsets[b+1])  (host array, B+1 ints)
 * pfn_weight    [C, F+2] fp32  (Linear.weight, pillar_modules.py:29)
 * pfn_scale/shift [C] fp32     eval-mode BatchNorm1d folded: y = relu(x.W^T * scale + shift)
 * Outputs (capacity rows = min(n_points, B*H*W), only the first *num_pillars are written):
 * pillar_indices    [cap, 3] int32 (b, y, x), ascending flat index b*H*W + y*W + x
 * pillar_features   [cap, C] fp32  max over the pillar's points, >= 0
 * point_pillar_index [n_points] int32 or NULL: row of each point's pillar, -1 if out of range
 * num_pillars       device int32[1]
 * occupancy_bits / word_rank: device views into the workspace that the sparse
 *               backbone reuses (returned through *occ_bits, *occ_rank; may be NULL).
 */
int pn_pillarize(const pnPillarGrid* g, const float* points, int64_t n_points,
                 const int32_t* batch_offsets_host, const float* pfn_weight,
                 const float* pfn_scale, const float* pfn_shift,
                 int32_t* pillar_indices, float* pillar_features,
                 int32_t* point_pillar_index, int32_t* num_pillars,
                 void* workspace, size_t workspace_bytes,
                 uint64_t** occ_bits, int32_t** occ_rank, pnStream stream);

/* ------------------------------------------------------------------------
 * (a10/a11) rotated-BEV NMS with on-device greedy selection, batched over
 * independent box sets (one per task x sample).
 *
 * Replaces iou3d_nms_cuda.nms_gpu  (det3d/ops/iou3d_nms/src/iou3d_nms_api.cpp:15,
 * iou3d_nms.cpp:113-159, nms_kernel iou3d_nms_kernel.cu:280-324) and the
 * `[:post_max_size]` slice of rotate_nms_pcdet (det3d/core/bbox/box_torch_ops.py:315-317).
 *
 * boxes      [num_sets, cap, 7] fp32, pcdet frame (x,y,z,dx,dy,dz,heading), each set
 *            sorted by score descending; cap <= PN_NMS_MAX_BOXES (the reference's nms_gpu has no limit;
 *            its callers pass at most nms_pre_max_size = 4096 boxes)
 * counts_dev device int32[num_sets] valid boxes per set (clamped to cap), or NULL to
 *            use the host value count_host for every set
 * thresh_dev device fp32[num_sets] IoU thresholds or NULL -> `thresh` for all; a negative threshold -t
 *            turns the set into circle NMS: squared centre distance <= t suppresses
 *            (det3d/core/utils/circle_nms_jit.py:5-28)
 * post_max_dev device int32[num_sets] or NULL -> `post_max` for all
 * keep       [num_sets, keep_stride] int32: positions (into the sorted set) of the
 *            kept boxes, ascending; num_keep device int32[num_sets]
 * ------------------------------------------------------------------------ */
#define PN_NMS_MAX_BOXES 16384
size_t pn_nms_workspace_bytes(int num_sets, int cap);
int pn_nms_rotated(const float* boxes, int num_sets, int cap, const int32_t* counts_dev,
                   int count_host, const float* thresh_dev, float thresh,
                   const int32_t* post_max_dev, int post_max, int keep_stride,
                   int32_t* keep, int32_t* num_keep, void* workspace,
                   size_t workspace_bytes, pnStream stream);
/* iou3d_nms_cuda.nms_normal_gpu (det3d/ops/iou3d_nms/src/iou3d_nms_api.cpp:16, iou3d_nms.cpp:162-215,
 * nms_normal_kernel iou3d_nms_kernel.cu:327-385): greedy NMS on the axis-aligned BEV IoU of boxes [n, 7] sorted by
 * score (heading ignored); keep [post_max] int32 positions, ascending; workspace: pn_nms_workspace_bytes(1, n). */
int pn_nms_normal(const float* boxes, int n, float thresh, int post_max, int32_t* keep, int32_t* num_keep,
                  void* workspace, size_t workspace_bytes, pnStream stream);

/* pn_nms_rotated without the workspace argument, for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference (whose nms_gpu calls cudaMalloc / cudaFree per call,
 * iou3d_nms.cpp:125-137).  The suppression bit matrix lives in a PN_NMS_STATIC_BYTES array inside
 * the library's device image (one copy per GPU, held until the process ends); the call allocates
 * and frees nothing.  Limit: the bit matrix alone must fit, num_sets * cap * ceil(cap / 64) * 8 <=
 * PN_NMS_STATIC_BYTES and cap <= 4096 (16 sets of 1024 boxes, or one set of 4096), otherwise PN_ENOSPC
 * (pn_nms_workspace_bytes is NOT the bound: it also counts the geometry block and the clip queue that
 * only pn_nms_rotated's workspace holds).  Without that queue the polygon clips run inside the
 * 64 x 64 tiles instead of one pair per lane over the whole chip: same results, slower than
 * pn_nms_rotated on sets with many overlapping pairs.  Every call on a GPU uses
 * that GPU's one array: calls that overlap on one GPU (two streams, two threads) race; calls
 * queued on one stream do not.  The first call on a device looks the array's address up
 * (hipGetSymbolAddress); make one call per device before capturing a graph. */
#define PN_NMS_STATIC_BYTES ((size_t)2 << 20)
int pn_nms_rotated_static(const float* boxes, int num_sets, int cap, const int32_t* counts_dev,
                          int count_host, const float* thresh_dev, float thresh,
                          const int32_t* post_max_dev, int post_max, int keep_stride,
                          int32_t* keep, int32_t* num_keep, pnStream stream);

/* iou3d_nms_cuda.boxes_overlap_bev_gpu / boxes_iou_bev_gpu / boxes_aligned_overlap_bev_gpu
 * (iou3d_nms_api.cpp:12-14, kernels iou3d_nms_kernel.cu:236-278).
 * out is [na, nb] (aligned = 0) or [na] (aligned = 1, na == nb); want_iou = 0 -> overlap area. */
int pn_boxes_pairwise_bev(const float* boxes_a, int na, const float* boxes_b, int nb,
                          int aligned, int want_iou, float* out, pnStream stream);

/* ------------------------------------------------------------------------
 * (a5/a6) 2-D sparse convolution on the bit/rank occupancy index.
 *
 * Stands in for spconv 2.x (third-party, not vendored by the reference) as used by
 * det3d/models/backbones/base.py:38-63 (conv2D3x3 -> SubMConv2d / SparseConv2d) and
 * det3d/models/backbones/PillarResNet.py:81-108,139 (SparseConv2d(3, 2, padding=1), .dense()).
 * Rows of every sparse tensor are sorted by flat index b*H*W + y*W + x; the active-row count
 * lives in device memory.
 * ------------------------------------------------------------------------ */

/* Rulebook of a 3x3 submanifold conv (stride 1, pad 1, dilation 1): nbr[k*cap + m] = row of the
 * active cell at offset k = ky*3+kx (dy = ky-1, dx = kx-1) from row m, or -1.  Shared by every
 * conv with the same indice_key. */
int pn_spconv_subm_rules(const int32_t* indices, const int32_t* count_dev, int cap,
                         int batch, int H, int W, const uint64_t* occ_bits,
                         const int32_t* occ_rank, int32_t* nbr, pnStream stream);

/* Active set + rulebook of a 3x3 / stride s / pad p sparse conv: builds the output level's
 * occupancy (out_bits, out_rank), sorted (b,y,x) rows, count, and nbr[k*out_cap + o] = input
 * row at window position k of output row o, or -1. */
size_t pn_spconv_down_scratch_bytes(int batch, int Ho, int Wo);
int pn_spconv_down_rules(const int32_t* in_indices, const int32_t* in_count_dev,
                         int in_cap, int batch, int H, int W, const uint64_t* in_bits,
                         const int32_t* in_rank, int stride, int pad, int Ho, int Wo,
                         int32_t* out_indices, int32_t* out_count_dev, int out_cap,
                         uint64_t* out_bits, int32_t* out_rank, int32_t* nbr,
                         void* scratch, size_t scratch_bytes, pnStream stream);

/* out[m, :] = act( (sum_k W[k] . in[nbr[k][m], :]) * scale + shift (+ residual[m, :]) )
 * in [in_rows, cin] (in_rows = rows allocated, bounds the LDS-DMA descriptor; 0 = unknown),
 * out/residual [cap, cout] of `dtype`; weight [9, cout, cin] of `dtype`
 * (spconv 2.x stores [cout, 3, 3, cin]); scale/shift fp32 [cout] (conv bias and eval-mode
 * BatchNorm1d folded; base.py:154-170); relu != 0 applies max(.,0).  fp32 accumulation. */
int pn_spconv_forward(int dtype, const void* in, int cin, long long in_rows, const int32_t* nbr,
                      int nbr_stride, const int32_t* count_dev, int cap,
                      const void* weight, const float* scale, const float* shift,
                      const void* residual, int relu, void* out, int cout, pnStream stream);

/* SparseConvTensor.dense(): zero-fills out and writes the active rows;
 * channels_last != 0 -> [B,H,W,out_ld] memory of which the first C channels are written (out_ld = 0
 * means C; a wider stride lets the rows land inside a concat buffer), else [B,C,H,W].
 * occ_bits / occ_rank (optional, channels-last only): the tensor's occupancy index; with it every
 * output cell is written once (row or zeros) and no memset pass is issued. */
int pn_sparse_to_dense(int dtype, const void* feats, const int32_t* indices,
                       const int32_t* count_dev, int cap, int batch, int H, int W, int C,
                       int channels_last, int out_ld, const uint64_t* occ_bits,
                       const int32_t* occ_rank, void* out, pnStream stream);

/* Occupancy index (bit per cell + per-word rank) of an existing sparse tensor's (b,y,x) rows --
 * what `spconv.SparseConvTensor(features, indices, spatial_shape, batch)`
 * (det3d/ops/pillar_ops/pillar_modules.py:74) needs before the first conv.  Rows must be in
 * ascending flat-index order; *active_cells_dev receives the number of distinct cells (equals the
 * row count iff there are no duplicates). */
size_t pn_occupancy_scratch_bytes(int batch, int H, int W);
int pn_occupancy_from_indices(const int32_t* indices, const int32_t* count_dev, int cap,
                              int batch, int H, int W, uint64_t* occ_bits,
                              int32_t* occ_rank, int32_t* active_cells_dev,
                              void* scratch, size_t scratch_bytes, pnStream stream);

/* ------------------------------------------------------------------------
 * (a9/a10) CenterHead.predict + post_processing + per-set rotated NMS, one call per frame batch.
 *
 * Replaces det3d/models/bbox_heads/center_head.py:217-413 (decode, class max, score / range
 * mask, task merge), det3d/core/bbox/box_torch_ops.py:296-359 (rotate_nms_pcdet,
 * rotate_class_nms_pcdet: rectified score, sort, top pre_max, [:post_max]) and
 * det3d/ops/iou3d_nms/iou3d_nms_utils.py:30-34 (to_pcdet).
 *
 * One pnDecodeGroup per NMS problem of a sample: a task, or a (task, class) pair for
 * use_multi_class_nms.  maps is that task's head output as [B, H*W, channels] fp32 with the
 * channel order  reg(2) height(1) dim(3) rot(2: sin, cos) [vel(2)] [iou(1)] hm(num_classes).
 * Output, per sample b: the kept boxes of its groups concatenated in group order --
 * out_boxes [B, out_cap, box_dim] (x,y,z,dx,dy,dz,[vx,vy,]heading), out_scores, out_labels
 * (int64 as the reference's label_preds, +label_offset), out_num[b].  *status bit 0: more cells than the 8192-slot candidate list holds tied
 * inside one score bin at the pre_max cut-off (excess dropped); bit 1: see pn_centerhead_detect_selected; bit 2: a
 * circle-NMS set had more cells above the score threshold than its pre_max (the reference ranks them all, here
 * the best pre_max).  Score ties are ordered by ascending cell index.
 * ------------------------------------------------------------------------ */
typedef struct {
  const float* maps;
  int32_t height, width, channels, num_classes;
  int32_t row_stride;      /* floats between consecutive cells of `maps` (0: = channels) */
  int32_t has_vel, has_iou;
  int32_t class_filter;    /* -1: all classes compete (rotate_nms_pcdet); k: only label k */
  int32_t label_offset;    /* center_head.py:337-343 */
  int32_t stride;          /* task stride (tasks[i].stride) */
  float score_threshold;   /* test_cfg.score_threshold */
  float rectifier;         /* test_cfg.rectifier: score^(1-r) * iou^r */
  int32_t use_rectify;     /* report the rectified score */
  float nms_threshold;     /* IoU threshold; a NEGATIVE value -t selects circle NMS: a kept box suppresses every
                              lower-scored one whose squared centre distance is <= t
                              (test_cfg.circular_nms / min_radius, det3d/core/utils/circle_nms_jit.py:5-28) */
  int32_t pre_max, post_max;
} pnDecodeGroup;

typedef struct {
  int32_t batch;
  float pillar_size, x_min, y_min;  /* fp32(pillar_size), fp32(point_cloud_range[0|1]) */
  int32_t has_range;
  float range[6];                   /* post_center_limit_range */
} pnDecodeCommon;

size_t pn_decode_workspace_bytes(const pnDecodeGroup* groups, int num_groups, int batch, int box_dim);
int pn_centerhead_detect(const pnDecodeGroup* groups, int num_groups,
                         const pnDecodeCommon* common, int box_dim, int out_cap,
                         float* out_boxes, float* out_scores, int64_t* out_labels,
                         int32_t* out_num, int32_t* status, void* workspace,
                         size_t workspace_bytes, pnStream stream);

/* The same detection path with the regression heads evaluated only where they are read
 * (det3d/models/bbox_heads/center_head.py:14-51 builds every head for every cell; predict,
 * center_head.py:353-413, reads them at <= nms_pre_max_size cells per task):
 *
 *   pn_centerhead_select           groups[g].maps = the task's HEAT MAP only, [B, H*W, row_stride] fp32 with
 *                                  channels == num_classes (has_vel / has_iou still describe the regression
 *                                  layout; rectifier must be 0).  Ranks the cells (score threshold, class
 *                                  filter, top pre_max, ties by ascending cell) and writes, per set (= group g,
 *                                  sample b; set = g*B + b), rulebooks for the caller's gather-GEMMs, with
 *                                  cap = pn_centerhead_select_capacity(...).  The rows of a set are the distinct
 *                                  cells of the 3x3 neighbourhoods of its selected cells, ascending (maps of up
 *                                  to 262 144 cells):
 *                                    row_counts [G*B]         rows in use (<= 9*cap), device
 *                                    nbr_tower [G*B][9][9*cap] shared-map row (b*H*W + cell) of each tap of the
 *                                                             first-layer tower conv at that row's cell, or -1
 *                                    nbr_out   [G*B][9][cap]  the nine tower rows of a selected cell, -1 where the
 *                                                             neighbour lies outside the map (zero padding)
 *   (caller)                       towers at the rows, output convs at the cap selected cells of every set ->
 *                                  regression rows [G][B*cap][row_stride] fp32 (reg, height, dim, rot, [vel], [iou])
 *   pn_centerhead_detect_selected  groups[g].maps = those rows (channels == 8 + 2*has_vel + has_iou), same
 *                                  workspace: box decode, the centre-range mask, NMS, task merge -- outputs as
 *                                  pn_centerhead_detect.  *status bit 1: a selected cell failed the range mask
 *                                  while its set held more candidates than pre_max -- the reference would have
 *                                  ranked another cell instead; redo the frame with pn_centerhead_detect.
 * The workspace (pn_decode_workspace_bytes) carries the selection from the first call to the second. */
int pn_centerhead_select_capacity(const pnDecodeGroup* groups, int num_groups);
int pn_centerhead_select(const pnDecodeGroup* groups, int num_groups, const pnDecodeCommon* common, int box_dim,
                         int32_t* nbr_tower, int32_t* nbr_out, int32_t* row_counts, void* workspace,
                         size_t workspace_bytes, pnStream stream);
int pn_centerhead_detect_selected(const pnDecodeGroup* groups, int num_groups, const pnDecodeCommon* common,
                                  int box_dim, int out_cap, float* out_boxes, float* out_scores,
                                  int64_t* out_labels, int32_t* out_num, int32_t* status, void* workspace,
                                  size_t workspace_bytes, pnStream stream);

/* ------------------------------------------------------------------------
 * LDS-tiled gather -> MFMA implicit GEMM, bf16 inputs / weights, fp32 accumulate:
 *   out[out_rows ? out_rows[m] : m, 0:cout] =
 *       act( (sum_{t < ntaps} W[t] . in[nbr[t*nbr_stride + m], :]) * scale + shift (+ residual[m, :]) )
 * The engine behind pn_spconv_forward (bf16) and behind the dense 3x3 / stride-2 / transposed
 * convolutions of conv5, the RPN necks and CenterHead when they run on channels-last rows
 * (det3d/models/backbones/PillarResNet.py:111-117, det3d/models/necks/rpn.py:147-207,
 * det3d/models/bbox_heads/center_head.py:14-51,101-127).
 * in [*, in_ld] bf16 rows of which the first cin channels are read (in_ld = 0: cin; offset `in`
 * to read a channel slice); in_rows = rows allocated behind `in` (bounds the buffer descriptor of
 * the LDS-DMA staging path; 0 selects register staging); nbr NULL = identity (ntaps must be 1); weight [ntaps, cout, cin] bf16;
 * out rows have stride out_ld elements (write a channel slice of a wider buffer by offsetting
 * `out`); out_fp32 != 0 stores fp32 instead of bf16; residual rows have stride res_ld.
 * ------------------------------------------------------------------------ */
int pn_gather_gemm(const void* in, int cin, int in_ld, long long in_rows, const int32_t* nbr, int nbr_stride, int ntaps,
                   const int32_t* count_dev, int rows, const void* weight, int cout,
                   const float* scale, const float* shift, const void* residual,
                   int res_ld, int relu, void* out, int out_ld, int out_fp32,
                   const int32_t* out_rows, pnStream stream);

/* Tile policy of the gather-GEMM launches (process-wide; returns the previous value; no counterpart in the
 * reference).  0 = a launch has the chip to itself (default).  1 = the caller keeps several frames in flight on
 * different streams: tiles are chosen so that two workgroups fit one CU and the frames' kernels share the chip.
 * Results are bit-identical under both.  A HIP graph keeps the tiles it was captured with. */
int pn_gg_set_tile_policy(int policy);

/* Grouped form of pn_gather_gemm: `groups` problems over the same rows / rulebook; group g reads the
 * channel slice starting at in + g*in_group_cols, uses weight + g*ntaps*cout*cin and
 * scale/shift + g*cout, and writes out + g*out_group_cols.  (CenterHead's per-task output convs,
 * det3d/models/bbox_heads/center_head.py:31-32,101-112, in one launch.)  out_rows (optional): output
 * row of site m for group g = out_rows[g*out_rows_group_stride + m] -- the four parity GEMMs of a
 * ConvTranspose2d(k2, s2) (det3d/models/necks/rpn.py:160-170) as one launch. */
int pn_gather_gemm_grouped(const void* in, int cin, int in_ld, long long in_rows, int in_group_cols,
                           const int32_t* nbr, int nbr_stride, int ntaps, int rows,
                           const void* weight, int cout, int groups, const float* scale,
                           const float* shift, int relu, void* out, int out_ld,
                           int out_group_cols, int out_fp32, const int32_t* out_rows,
                           long long out_rows_group_stride, pnStream stream);

/* Grouped form whose groups also differ in rulebook and row block (the lazily evaluated CenterHead towers:
 * one block of gathered rows per task): group g gathers through nbr + g*nbr_group_stride from
 * in + g*in_group_stride (elements; 0: all groups gather from the same rows), uses weight + g*ntaps*cout*cin,
 * scale/shift + g*cout, and writes out + g*out_group_stride (elements). */
int pn_gather_gemm_blocks(const void* in, int cin, int in_ld, long long in_rows, long long in_group_stride,
                          const int32_t* nbr, int nbr_stride, long long nbr_group_stride, int ntaps,
                          const int32_t* counts_dev /* [groups] rows in use per group, or NULL: rows */, int rows,
                          const void* weight, int cout, int groups, const float* scale, const float* shift,
                          int relu, void* out, int out_ld, long long out_group_stride, int out_fp32,
                          pnStream stream);

/* Spatial half of a 3x3 conv whose per-tap channel products z = x . [W_0 | ... | W_8] were computed
 * first (as a 1-tap GEMM): out[p, c] = bias[c] + sum_tap z[nbr[tap][p], tap*channels + c], summed and
 * stored in fp32; z is fp32 or bf16 (z_dtype = PN_DTYPE_*; strides in elements).
 * Grouped: group g uses z + g*z_group_stride, bias + g*channels, out + g*out_group_stride. */
int pn_tap_sum(const void* z, int z_dtype, int z_ld, long long z_group_stride, const int32_t* nbr,
               int nbr_stride, int rows, int channels, int groups, const float* bias,
               float* out, int out_ld, long long out_group_stride, pnStream stream);

/* pn_tap_sum with one rulebook per group (nbr + g*nbr_group_stride). */
int pn_tap_sum_blocks(const void* z, int z_dtype, int z_ld, long long z_group_stride, const int32_t* nbr,
                      int nbr_stride, long long nbr_group_stride, int rows, int channels, int groups,
                      const float* bias, float* out, int out_ld, long long out_group_stride, pnStream stream);

/* ------------------------------------------------------------------------
 * Unfused / training-path operators with the reference's pybind contracts
 * (det3d/ops/pillar_ops/src/pillar_api.cpp:10-21; all tensors caller-allocated, on the device).
 * ------------------------------------------------------------------------ */

/* pillar_cuda.create_point_pillar_index_stack_wrapper (pillar_ops.cpp:15-36): pts_xy i32[N,2],
 * pts_batch_cnt i32[B] (device), pillars_mask u8/bool[B,H,W] pre-zeroed, point_pillar_index
 * i32[N] pre-filled with -1. */
int pn_create_point_pillar_index_stack(const int32_t* pts_xy, const int32_t* pts_batch_cnt, int N,
                                       int B, int H, int W, uint8_t* pillars_mask,
                                       int32_t* point_pillar_index, pnStream stream);
/* pillar_cuda.create_pillar_indices_wrapper (pillar_ops.cpp:39-55): pillars_position i32[B,H,W]
 * (-1 = empty) -> pillar_indices i32[M,3]. */
int pn_create_pillar_indices(const int32_t* pillars_position, int B, int H, int W,
                             int32_t* pillar_indices, pnStream stream);
/* pillar_cuda.gather_indice_wrapper / gather_feature_wrapper (group_ops.cpp:4-38). */
int pn_gather_indice(const int32_t* index, int L, const int32_t* indices, int32_t* outs, pnStream stream);
int pn_gather_feature(const int32_t* index, int L, const float* features, int C, float* outs,
                      pnStream stream);
/* pillar_cuda.gather_feature_grad_wrapper (group_ops.cpp:41-57; kernel group_ops_gpu.cu:35-48):
 * grad_features[index[l], :] += grad_outs[l, :] (fp32 atomics; grad_features is accumulated into, the caller
 * zeroes it).  Argument order of the C++ binding; the reference's Python caller (group_utils.py:28-35) passes
 * (index, grad_outs, grad_features) and is never reached, because its one input is detached. */
int pn_gather_feature_grad(const float* grad_outs, const int32_t* index, int L, int C, float* grad_features,
                           pnStream stream);
/* pillar_cuda.scatter_max_wrapper (scatter_ops.cpp:7-24): out[m,c] = max(0, max_{p: index[p]=m}
 * src[p,c]); arg may be NULL.  One pass, deterministic.  workspace: pn_scatter_max_workspace_bytes(M, C).
 * `arg` is NOT the reference's encoding: the reference's second pass (scatter_ops_gpu.cu:24-36) stores the
 * flat element index p*C + c of whichever maximal point wrote last, zero-valued maxima included; here
 * arg[m,c] = the POINT index p of the lowest maximal point, and -1 where out stayed 0 (such cells get no
 * gradient either way: the feature fed to scatter_max is post-ReLU).  It is consumed by pn_scatter_max_grad
 * below, which expects this encoding; a caller that keeps the reference's own scatter_max_grad kernel must
 * convert (arg * C + c). */
size_t pn_scatter_max_workspace_bytes(int M, int C);
int pn_scatter_max(const int32_t* index, const float* src, int L, int C, int M, int32_t* arg,
                   float* out, void* workspace, size_t workspace_bytes, pnStream stream);
/* pillar_cuda.scatter_max_grad_wrapper (scatter_ops.cpp:26-41): grad_src (pre-zeroed [L,C])
 * receives grad_out[m,c] at row arg[m,c]. */
int pn_scatter_max_grad(const float* grad_out, const int32_t* arg, int M, int C, float* grad_src,
                        pnStream stream);
/* Inverse rulebook for the sparse-conv data gradient: inv[k*inv_stride + j] = the output row m
 * with nbr[k*nbr_stride + m] == j, or -1 (inv is fully initialised here). */
int pn_spconv_invert_rules(const int32_t* nbr, int nbr_stride, const int32_t* out_count_dev,
                           int out_cap, int32_t* inv, int inv_stride, pnStream stream);
/* Weight gradient of a rulebook convolution, fp32:
 *   dw[co][tap][ci] = sum_{m < rows} grad_out[m][co] * x[nbr[tap*nbr_stride + m]][ci]      (nbr < 0: no term)
 * which is the layout of an spconv 2.x weight [cout, 3, 3, cin] for ntaps = 9.  Replaces the weight-gradient
 * half of spconv's backward (third-party spconv-cu114, reached through spconv.pytorch.SubMConv2d / SparseConv2d
 * at det3d/models/backbones/base.py:9-36; the reference holds no source for it).  nbr NULL = identity
 * (ntaps must be 1: a plain X^T G product).  cin, cout multiples of 32; x_ld / g_ld = row strides in floats
 * (0: cin / cout); count_dev optional device row count.  Partial sums per site chunk live in `workspace`
 * (pn_spconv_wgrad_workspace_bytes) and are added in a fixed order: results are bit-reproducible.  Widths that are
 * multiples of 64 with 16-byte aligned rows (pointers and strides) take the kernel that stages whole rows through
 * LDS (about twice as fast on the 128- and 256-wide levels); anything else the per-tile kernel. */
size_t pn_spconv_wgrad_workspace_bytes(int rows, int cin, int cout, int ntaps);
int pn_spconv_wgrad(const float* x, int cin, int x_ld, const int32_t* nbr, int nbr_stride, int ntaps,
                    const int32_t* count_dev, int rows, const float* grad_out, int cout, int g_ld,
                    float* dw, void* workspace, size_t workspace_bytes, pnStream stream);

/* ------------------------------------------------------------------------
 * RoI pillar pooling of Pillar R-CNN's second stage: a grid x grid lattice over every rotated RoI,
 * bilinearly interpolated from a channels-last BEV map.
 * Replaces BEVStrideFeature.get_pooling_points + interpolate_from_bev_features
 * (det3d/models/second_stage/bev_interpolation.py:236-268), center_to_grid_box2d
 * (det3d/core/bbox/box_torch_ops.py:220-251) and bilinear_interpolate_torch
 * (det3d/core/utils/center_utils.py:91-120).
 * rois [batch, rois_per_sample, roi_dim] fp32 (x,y,z,dx,dy,dz,...,heading = last element);
 * map [batch, H*W, map_ld] of `dtype`, first C channels used; cell_size = out_stride * pillar_size;
 * out [batch, rois_per_sample, grid*grid, C] fp32; points (optional) [.., grid*grid, 2] fp32.
 * ------------------------------------------------------------------------ */
int pn_roi_grid_pool(int dtype, const float* rois, int roi_dim, int batch, int rois_per_sample,
                     int grid, const void* map, int map_ld, int H, int W, int C, float x_min,
                     float y_min, float cell_size, float* out, float* points, pnStream stream);

/* Gradient of pn_roi_grid_pool w.r.t. the map (training of the second stage): grad_out [batch, rois_per_sample,
 * grid*grid, C] fp32 -> grad_map [batch, H*W, C] fp32, which must be ZERO on entry (fp32 atomic adds into the
 * four bilinear corners of every grid point).  The reference has no kernel here: autograd differentiates the
 * gathers of bilinear_interpolate_torch (det3d/core/utils/center_utils.py:91-120). */
int pn_roi_grid_pool_grad(const float* rois, int roi_dim, int batch, int rois_per_sample, int grid,
                          const float* grad_out, int H, int W, int C, float x_min, float y_min,
                          float cell_size, float* grad_map, pnStream stream);

/* ------------------------------------------------------------------------
 * points_in_boxes_2d_gpu of the reference's roiaware_pool3d extension (the PointHead target assigner of
 * Pillar R-CNN): det3d/ops/roiaware_pool3d/src/roiaware_pool3d.cpp (points_in_boxes_2d_gpu),
 * roiaware_pool3d_kernel.cu:39-50,354-377.  boxes [batch, boxes_num, 7] fp32 in pcdet convention
 * (x, y, z, dx, dy, dz, heading), pts [batch, pts_num, 2]; box_idx_of_points [batch, pts_num] int32 =
 * index of the first box whose BEV rectangle (margin 1e-5) holds the point, else -1 (always written).
 * ------------------------------------------------------------------------ */
int pn_points_in_boxes_2d(const float* boxes, const float* pts, int batch, int boxes_num, int pts_num,
                          int32_t* box_idx_of_points, pnStream stream);

/* ------------------------------------------------------------------------
 * Multi-sweep merge (the data step in front of the reader).
 * Replaces the nuScenes branch of LoadPointCloudFromFile.__call__ with read_sweep / remove_close
 * (det3d/datasets/pipelines/loading.py:38-61,103-140) and its Waymo branch with read_single_waymo /
 * read_single_waymo_sweep (:63-94,142-175) for sweeps already resident in HBM:
 *   raw [n_total, raw_dim] fp32 = the .bin files back to back (key sweep first); sweep s owns rows
 *   [sweep_start[s], sweep_start[s+1]) (HOST array of n_sweeps+1 ints).  Per sweep (device arrays):
 *   transform[s][16] float64 row-major 4x4 (used where has_transform[s]), time_lag[s] fp32,
 *   filter_close[s] (1: drop points with |x| < radius and |y| < radius, tested before the transform).
 *   tanh_column: -1, or the feature column (3 <= c < use_dim) stored as tanhf(value) -- Waymo's intensity
 *   normalisation (loading.py:67-68,79-80).
 * out [<= n_total, use_dim + 1] fp32: (x', y', z', raw[3..use_dim), time lag), reference row order
 * (a stable compaction); out_count: device int32.  Workspace: pn_merge_sweeps_workspace_bytes(n_total).
 * ------------------------------------------------------------------------ */
size_t pn_merge_sweeps_workspace_bytes(int n_points);
int pn_merge_sweeps(const float* raw, int raw_dim, int use_dim, const int32_t* sweep_start_host,
                    int n_sweeps, const double* transform, const uint8_t* has_transform,
                    const float* time_lag, const uint8_t* filter_close, float radius, int tanh_column,
                    float* out, int32_t* out_count, void* workspace, size_t workspace_bytes, pnStream stream);

/* ------------------------------------------------------------------------
 * Host-side hard voxelizer: the `Voxelization(cfg=voxel_generator)` pipeline step of the legacy configs,
 * det3d/ops/point_cloud/point_cloud_ops.py:112-184 (points_to_voxel; numba kernels :8-109).  CPU, as in the
 * reference (dataloader workers): HOST pointers.  points [n, feats] fp32; grid_map int32 per voxel of the grid
 * (z, y, x order when reverse_index), all -1 on entry; voxels [max_voxels, max_points, feats] and num_points
 * zeroed on entry; coors [max_voxels, 3].  Returns the number of voxels (>= 0) or PN_EINVAL.
 * ------------------------------------------------------------------------ */
int pn_points_to_voxel(const float* points, long long n, int feats, const float* voxel_size, const float* range,
                       int reverse_index, int max_points, int max_voxels, int32_t* grid_map, float* voxels,
                       int32_t* coors, int32_t* num_points);

#ifdef __cplusplus
}
#endif
#endif /* PILLARNET_HIP_H_ */
