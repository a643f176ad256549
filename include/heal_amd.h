/*
 * heal_amd.h -- C ABI of libheal_amd.so: the MI355X (gfx950) kernels of HEAL's per-frame
 * perception hot path.
 *
 * The reference (yifanlu0227/HEAL) is pure Python; it has no FFI of its own.  Each entry point
 * below replaces the arithmetic of one reference function (cited as file:line, paths relative to
 * the reference root) and is what a ctypes stub on the reference side would bind.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous;
 *   - `ws` is caller-owned device scratch of at least heal_*_workspace(...) bytes, 256-B aligned;
 *   - return value: 0 on success, non-zero on error; heal_last_error() gives the message
 *     (thread-local).  No call allocates, frees or synchronises.
 *   - all floating point is fp32 unless stated, indices int32, layouts are C-contiguous.
 */
#ifndef HEAL_AMD_H
#define HEAL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HEAL_AMD_ABI_VERSION 7   /* 6 (round 6): + heal_pfn_pillars / heal_pillar_canvas / heal_pillar_stem_block; the measured-negative entry
                                   points moved to include/heal_amd_experimental.h (HEAL_BUILD_EXPERIMENTAL=1 builds only) */

int heal_abi_version(void);
const char* heal_last_error(void);

/* heal_fill_bytes: dst[0 .. bytes) <- byte, as a KERNEL (dst and bytes multiples of 4).  Every table / counter initialisation inside this
 * library goes through it; hipMemsetAsync is not used: captured into a HIP graph it becomes a memset node, and the round-4 memory fault
 * was such a node writing 00 FF FF .. FF for every 16 bytes of a 0xFF fill (profiles/r05_k1_memset_node_dump.txt).  Exported so that
 * callers that prepare buffers for this library inside a captured region (and the tests) can use the same primitive.  No reference
 * counterpart: torch's `x.fill_()` / `torch.zeros` play this role around opencood's CUDA extensions.                                     */
int heal_fill_bytes(void* dst, int byte, size_t bytes, void* stream);
/* Measurement hook: arm a pair of hipEvent_t (created with timing enabled) for THIS thread; the next launch of an entry point
 * that supports it -- heal_bev_pool_scatter, heal_bev_stem_block -- stamps them with the kernel's own begin / end timestamps
 * (hipExtLaunchKernelGGL: the interval a rocprofv3 kernel trace reports, without the dispatch / marker latencies an event pair
 * recorded around the launch includes).  One shot; (NULL, NULL) disarms.  Not capturable in a HIP graph.                      */
int heal_next_launch_events(void* start_event, void* stop_event);

/* ------------------------------------------------------------------------------------------------
 * K1  hard voxelisation (first-come, input-order semantics of spconv's point->voxel generator).
 * Replaces: opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:62-85 (preprocess ->
 *           spconv VoxelGeneratorV2.generate / Point2VoxelCPU3d.point_to_voxel) and the batch-index
 *           prepend of collate_batch_list (:110-147).
 *   points      [n_points,4] f32 (x,y,z,intensity)
 *   range       host, 6 floats (xmin,ymin,zmin,xmax,ymax,zmax); voxel_size host, 3 floats
 *   voxels      [cap,max_points,4] f32, rows [0,M) written (zero padded); cap = min(n_points,max_voxels)
 *   coords      [cap,4] i32 (batch_idx,z,y,x), rows [0,M)
 *   num_points  [cap] i32
 *   n_voxels    [1] i32  <- M
 *   row_offset / row_offset_next  device i32 or NULL: collate_batch_list without a host round trip -- this agent's
 *               rows go to [*row_offset, *row_offset + M) of voxels/coords/num_points (buffers shared by the agents
 *               of a modality) and *row_offset_next <- *row_offset + M feeds the next agent's call.
 *   ws / tables_clean  round 6: the per-cell table inside the workspace is SELF-CLEANING -- the chain's last kernel resets every record it
 *               used -- so only the FIRST call on a workspace has to initialise it: tables_clean = 0 "contents unknown" (the call fills the
 *               tables first: two more launches), 1 "the previous call on this workspace used the same (n_points, max_points, max_voxels)
 *               and ran to completion" (no fill; three kernels).  A call with other sizes carves the workspace differently: pass 0 again.
 * -----------------------------------------------------------------------------------------------*/
/* agents_x_cells (size queries): agents x grid cells of the call, or 0.  Grids of up to 2^21 cells in total (every PointPillars grid: 3 agents
 *   at 512 x 512 = 786 432) then get a per-cell table instead of a hash grid -- no key, no probe; free per call because the table cleans itself. */
size_t heal_voxelize_workspace(int n_points, int max_points, int max_voxels, long long agents_x_cells);
int heal_voxelize(const float* points, int n_points,
                  const float* range_host, const float* voxel_size_host,
                  int max_points, int max_voxels, int batch_idx,
                  float* voxels, int32_t* coords, int32_t* num_points, int32_t* n_voxels,
                  const int32_t* row_offset, int32_t* row_offset_next,
                  void* ws, size_t ws_bytes, int tables_clean, void* stream);

/* heal_voxelize_batch: K1 for every agent of a modality in ONE launch chain (three kernels on a clean workspace, as the single-cloud
 *   form; at most 4 M points per call): `points` holds the agents' clouds back to back, point_offsets_host [n_agents+1] (host)
 *   the boundaries.  Outputs are the collated buffers of collate_batch_list: rows of agent b at
 *   [row_offsets[b], row_offsets[b+1]) with coords (b,z,y,x); row_offsets [n_agents+1] i32 DEVICE.  Buffers need
 *   sum_b min(n_b, max_voxels) rows.  Same semantics per agent as heal_voxelize (first-come order, both caps).  <= 16
 *   agents per call.                                                                                              */
size_t heal_voxelize_batch_workspace(int n_points_total, int n_agents, int max_points, int max_voxels, long long agents_x_cells);
int heal_voxelize_batch(const float* points, const int32_t* point_offsets_host, int n_agents,
                        const float* range_host, const float* voxel_size_host, int max_points, int max_voxels,
                        float* voxels, int32_t* coords, int32_t* num_points, int32_t* row_offsets,
                        void* ws, size_t ws_bytes, int tables_clean, void* stream);

/* heal_mask_points: the point filters the dataset applies right before the voxeliser, on the device.
 * Replaces: opencood/utils/pcd_utils.py:41-67 (mask_points_by_range: strict > / < on x, y, z) and :70-88
 *           (mask_ego_points: drop -1.95 <= x <= 2.95 and -1.1 <= y <= 1.1), called from the datasets'
 *           get_item_single_car before SpVoxelPreprocessor.preprocess.
 *   points [n_points,4] f32 -> out [n_points,4] f32 (may alias points): a kept point is copied, a dropped point
 *   becomes four NaNs.  The cloud keeps its length and order -- no compaction, no host round trip -- and the
 *   voxeliser drops NaN points, so heal_voxelize(out) equals the reference's preprocess(points[mask]) exactly
 *   (first-come voxel order and both caps only see surviving points).  range_host: 6 floats or NULL (no range
 *   filter); mask_ego: 0 | 1.  Comparisons in fp32 against the fp32-rounded bounds, like numpy on a float32 array. */
int heal_mask_points(const float* points, int n_points, const float* range_host, int mask_ego, float* out,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * K2  fused pillar feature net + scatter to the dense BEV canvas.
 * Replaces: opencood/models/sub_modules/pillar_vfe.py:105-155 (PillarVFE.forward),
 *           :31-53 (PFNLayer.forward, single last layer, use_norm, eval-mode BatchNorm1d),
 *           opencood/models/sub_modules/point_pillar_scatter.py:19-76 (PointPillarScatter.forward).
 *   voxels [M,P,4], coords [M,4] (b,z,y,x), num_points [M]
 *   n_voxels_dev: optional device int; when non-NULL the kernels use min(*n_voxels_dev, M) pillars
 *   weight [C,10] (nn.Linear weight), bn_scale[C] = gamma/sqrt(var+eps), bn_shift[C] = beta-mean*scale
 *   canvas [n_agents,C,ny,nx] f32 -- every element is written (zeros where no pillar)
 *   pillar_feat (optional, may be NULL) [M,C] f32 gets the PFN output (pillar_vfe.py:153)
 *   C must be 64, P <= 64.
 * -----------------------------------------------------------------------------------------------*/
size_t heal_pfn_scatter_workspace(int n_voxels, int n_agents, int ny, int nx, int channels);
int heal_pfn_scatter(const float* voxels, const int32_t* coords, const int32_t* num_points,
                     int n_voxels, const int32_t* n_voxels_dev, int max_points,
                     const float* weight, const float* bn_scale, const float* bn_shift, int channels,
                     float vx, float vy, float vz, float x_offset, float y_offset, float z_offset,
                     int n_agents, int ny, int nx,
                     float* canvas, float* pillar_feat,
                     void* ws, size_t ws_bytes, void* stream);

/* Round 6: K2 without the dense canvas, for consumers that read the pillars themselves.
 * heal_pfn_pillars: PillarVFE + PFNLayer (pillar_vfe.py:31-155) only -> pillar_feat [n_voxels, 64] and cell_map [n_agents, ny, nx]
 *   (int32: the pillar row that PointPillarScatter.forward, point_pillar_scatter.py:58-65, would copy into that cell -- the highest
 *   row when several rows name one cell, -1 for an empty cell).  Same arguments as heal_pfn_scatter otherwise; no workspace.
 * heal_pillar_canvas: the reference's canvas [n_agents, channels, ny, nx] from (cell_map, pillar_feat): what heal_pfn_scatter writes.
 * heal_pillar_stem_block: the first BasicBlock convolutions of the PointPillars ResNetBEVBackbone (base_bev_backbone_resnet.py:88-109,
 *   resblock.py:18-64) straight from the pillars: out_main = relu(conv3x3/2 pad 1 (canvas, W1) + b1), out_identity = conv1x1/2 (canvas,
 *   Wd) + bd, both [n_agents, 64, ceil(ny/2), ceil(nx/2)], BatchNorms folded by the caller; the canvas is never written (it is 96 % zeros:
 *   only the output pixels that see a pillar are multiplied, in groups of 16 on the 16x16x4 fp32 MFMA).  channels must be 64; biases may be NULL.
 *   weight_layout 0 (production): lane fragments w_main [9][4][64][16], w_down [4][64][16] with frag[tap][w][16 lk + ln][ks] =
 *   W[16 w + ln][16 lk + ks][tap] (heal_amd.ops.pillar_stem_fragments); weight_layout 1 (the round's first version, 8 x 8 tile x tap
 *   skipping on 32x32x2 MFMA, kept for A/B): w_main [9][1][64][64] (tap = ky*3 + kx, cout, cin), w_down [1][64][64] (ops.stem_fragments). */
int heal_pfn_pillars(const float* voxels, const int32_t* coords, const int32_t* num_points, int n_voxels,
                     const int32_t* n_voxels_dev, int max_points, const float* weight, const float* bn_scale,
                     const float* bn_shift, int channels, float vx, float vy, float vz, float x_offset, float y_offset,
                     float z_offset, int n_agents, int ny, int nx, float* pillar_feat, int32_t* cell_map, void* stream);
int heal_pillar_canvas(const int32_t* cell_map, const float* pillar_feat, int n_agents, int channels, int ny, int nx,
                       float* canvas, void* stream);
int heal_pillar_stem_block(const float* pillar_feat, const int32_t* cell_map, int n_agents, int channels, int ny, int nx,
                           const float* w_main, const float* b_main, const float* w_down, const float* b_down,
                           int weight_layout, float* out_main, float* out_identity, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K5  warp to ego + occupancy-softmax weighted fusion over agents (one pyramid level, one scene).
 * Replaces: opencood/models/fuse_modules/pyramid_fuse.py:17-63 (weighted_fuse) including
 *           warp_affine_simple (opencood/models/sub_modules/torch_transformation_utils.py:323-332:
 *           F.affine_grid + F.grid_sample bilinear / zeros / align_corners=False), the score
 *           construction sigmoid(occ)+1e-4 (pyramid_fuse.py:145) and the eval-mode camera crop
 *           mask (pyramid_fuse.py:147-162).
 *   feats  [n_agents,C,H,W]; occ [n_agents,1,H,W] (logits of single_head_i)
 *   affine_host: host, n_agents*6 doubles = rows of affine_matrix[b][0, a] (2x3) from
 *                normalize_pairwise_tfm (opencood/utils/transformation_utils.py:68-92); may be NULL
 *                when affine_dev is given
 *   affine_dev:  the same rows in DEVICE memory, or NULL.  When non-NULL the kernel reads the poses
 *                at run time (the reference keeps pairwise_t_matrix on the device too,
 *                train_utils.to_device): a HIP graph captured around the call then follows whatever
 *                the buffer holds at replay time instead of freezing the captured frame's poses
 *   grid_f64: non-zero -> sampling grid computed in fp64 then rounded to fp32 (the reference's
 *             behaviour when pairwise_t_matrix is float64), zero -> fp32 throughout
 *   crop_host: host, n_agents*4 ints (h0,h1,w0,w1) rectangle where the camera score is KEPT
 *              (outside -> score 0); h1<=h0 means "no mask" (lidar agent); may be NULL
 *   out    [C,H,W]
 * -----------------------------------------------------------------------------------------------*/
int heal_warp_fuse(const float* feats, const float* occ, int n_agents, int channels, int H, int W,
                   const double* affine_host, const double* affine_dev, int grid_f64,
                   const int32_t* crop_host, float* out, void* stream);

/* heal_warp_fuse_levels: heal_warp_fuse for ALL levels of the pyramid in one launch (pyramid_fuse.py:104-168 fuses the three
 *   levels back to back), the source footprint of every 16 x 16 ego tile staged through LDS as row segments instead of gathered
 *   word by word.  Level l: feats_host[l] [n_agents, C_l, H_l, W_l], occ_host[l] [n_agents, 1, H_l, W_l], out_host[l] [C_l, H_l, W_l]
 *   (HOST arrays of device pointers); the affine rows are shared by the levels (normalised coordinates); crop_host
 *   [n_levels][n_agents][4] or NULL.  Same arithmetic, operation for operation, as heal_warp_fuse: bit-identical results.  */
int heal_warp_fuse_levels(int n_levels, const float* const* feats_host, const float* const* occ_host, int n_agents,
                          const int32_t* channels_host, const int32_t* h_host, const int32_t* w_host,
                          const double* affine_host, const double* affine_dev, int grid_f64, const int32_t* crop_host,
                          float* const* out_host, void* stream);

/* heal_warp_fuse_backward: gradient of heal_warp_fuse with respect to the agents' maps and occupancy logits (training; the
 *   autograd of warp_affine_simple x 2 + masked softmax + weighted sum, pyramid_fuse.py:17-63,145-162).  Arguments as
 *   heal_warp_fuse plus grad_out [C,H,W]; grad_feats [n_agents,C,H,W] and grad_occ [n_agents,1,H,W] must be ZERO on entry
 *   (bilinear taps of several ego pixels add into one source pixel: fp32 atomics).                                   */
int heal_warp_fuse_backward(const float* feats, const float* occ, int n_agents, int channels, int H, int W,
                            const double* affine_host, const double* affine_dev, int grid_f64, const int32_t* crop_host,
                            const float* grad_out, float* grad_feats, float* grad_occ, void* stream);

/* Same operator split for agent-sharded execution: warp ONE agent's features and score
 * into the ego frame (rank-local, before the all-gather) ...                                      */
int heal_warp_agent(const float* feat, const float* occ, int channels, int H, int W,
                    const double* affine_host, const double* affine_dev, int grid_f64,
                    const int32_t* crop_host, float* feat_ego, float* score_ego, void* stream);
/* heal_warp_agents_pm: warp every agent of a scene into the ego frame and write the maps TOKEN-MAJOR:
 *   feats [n_agents, C, H, W] -> out [n_agents, H, W, C].  Replaces warp_affine_simple (torch_transformation_utils.py:323-332)
 *   + `x.permute(0, 2, 3, 1)` at the entry of V2XViTFusion (opencood/models/fuse_modules/fusion_in_one.py:352-358): one pass
 *   instead of a warp per agent, a stack and a layout copy.  C % 4 == 0; affine rows as for heal_warp_fuse.                */
int heal_warp_agents_pm(const float* feats, int n_agents, int channels, int H, int W, const double* affine_host,
                        const double* affine_dev, int grid_f64, float* out, void* stream);
/* ... and fuse already-warped stacks (after the all-gather): -inf mask, softmax over agents, sum. */
int heal_fuse_warped(const float* feats_ego, const float* scores_ego, int n_agents, int channels,
                     int H, int W, float* out, void* stream);
/* heal_fuse_warped_rows: the same fusion reading every agent's maps IN PLACE from one buffer -- agent a's [C,H,W] features at
 *   base + feat_offsets_host[a] floats, its [H,W] scores at base + score_offsets_host[a] (multiples of 4 floats).  The
 *   agent-sharded runner points it at the rows of the exchange buffer, so nothing is re-packed after the gather. */
int heal_fuse_warped_rows(const float* base, const int64_t* feat_offsets_host, const int64_t* score_offsets_host,
                          int n_agents, int channels, int H, int W, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K8  box decode + filters + rotated NMS.
 * Replaces: opencood/data_utils/post_processor/voxel_postprocessor.py:245-405 (post_process),
 *           :407-453 (delta_to_boxes3d), opencood/utils/box_utils.py:152-204 (boxes_to_corners_3d),
 *           :278-316 (project_box3d), :840-890 (remove_large_pred_bbx, remove_bbx_abnormal_z),
 *           :693-738 (nms_rotated, shapely IoU via common_utils.py:230-270),
 *           :384-421 (mask_boxes_outside_range_numpy).
 *   cls [A,H,W], reg [7A,H,W], dir [num_bins*A,H,W] (may be NULL), anchors [H,W,A,7] f32 (x,y,z,h,w,l,yaw)
 *   tfm_host: host 16 floats row-major 4x4 (ego <- cav), order 'hwl'
 *   gt_range_host: host 6 floats
 *   out_corners [max_out,8,3] f32, out_scores [max_out] f32, out_count [1] i32 (<= max_out, max_out>=nms_top)
 * -----------------------------------------------------------------------------------------------*/
size_t heal_decode_nms_workspace(int anchors_total, int nms_top);
int heal_decode_nms(const float* cls, const float* reg, const float* dir, const float* anchors,
                    int H, int W, int anchor_num, int num_bins,
                    float score_thr, float dir_offset, float nms_thr, int nms_top,
                    const float* tfm_host, const float* gt_range_host,
                    float* out_corners, float* out_scores, int32_t* out_count, int max_out,
                    void* ws, size_t ws_bytes, void* stream);

/* Pairwise rotated IoU of convex quads given as 4 (x,y) fp32 corners; fp64 geometry, fp32 result.
 * Same arithmetic as the NMS above (common_utils.py:230-251 compute_iou).  a [n,4,2], b [m,4,2],
 * iou [n,m].                                                                                       */
int heal_quad_iou(const float* a, int n, const float* b, int m, float* iou, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K4  Lift-Splat frustum -> BEV pooling, fused with the depth softmax and the outer product.
 * Replaces: opencood/models/heter_encoders.py:125-147 (get_geometry), :161-217 (voxel_pooling),
 *           opencood/utils/camera_utils.py:220-236 (QuickCumsum.forward) and the
 *           softmax(depth) (x) feat outer product of lss_submodule.py:129-134.
 *   depth_logit [n_agents*n_cams,D,fH,fW]; feat [n_agents*n_cams,C,fH,fW]
 *   frustum [D,fH,fW,3] f32 (create_frustum, heter_encoders.py:110-123)
 *   cam_mats [n_agents*n_cams,27] f32 DEVICE: combine = rots @ inv(intrins) (9), inv(post_rots) (9),
 *             post_trans (3), trans (3), pad (3)   -- the 3x3 algebra is a handful of tiny device ops
 *             upstream, so no host round trip is needed in the middle of the forward pass
 *   dx,bx host 3 floats each, nx host 3 ints (gen_dx_bx, camera_utils.py:129-134)
 *   out [n_agents, C*nz, ny, nx] f32, every element written
 *   heal_bev_pool takes the reference's NCHW tensors and runs the bit-reproducible radix-sort pipeline (stable sort of
 *   the 590 k lifted points by cell, segmented reduction in point order): any shape, ~10 launches.
 *   heal_bev_pool_pm is the pixel-major path the models run: `head` [n_agents*n_cams, fH*fW, head_stride] is the PIXEL-MAJOR
 *   output of the fused image_head | depth_head convolution (heal_conv1x1 out_pixel_major): per pixel C image features
 *   followed by D depth logits (head_stride >= C + D floats); fH <= 64, D <= 64, a separable frustum (frustum[d][v][u] =
 *   (xs[u], ys[v], ds[d]), what create_frustum builds).  Within one image column (camera, u, depth bin) the summation
 *   order is fixed; across columns that feed the same cell it adds with fp32 hardware atomics, so a result can differ in
 *   the last bit from call to call (like the reference's unstable argsort + cumsum difference).
 *   It is two steps that can also be called separately:
 *     heal_bev_pool_scatter   ONE launch: lift + splat into the SPARSE PIXEL-MAJOR BEV map kept in `ws`: rows[cell][C] for the
 *                             cells any point fell into + a generation-tagged flag per cell (a level rig touches <= n_cams*fW*D
 *                             of the ny*nx cells: 19 % at BASELINE size);
 *     then EXACTLY ONE consumer of that map, before the next scatter into the same `ws`:
 *     heal_bev_pool_emit      the dense [n_agents, C*nz, ny, nx] tensor of the reference (every element written), or
 *     heal_bev_stem_block     the first BasicBlock of the camera ResNetBEVBackbone read straight from the sparse map
 *                             (base_bev_backbone_resnet.py:88-109, resblock.py:18-64 with stride 2 and a 1x1 downsample):
 *                             out_main = relu(conv3x3_s2(x, W1) + b1), out_identity = conv1x1_s2(x, Wd) + bd, both
 *                             [n_agents, 64, ny/2, nx/2] NCHW; BN folded into (W, b) by the caller; nz = 1, C % 32 == 0;
 *                             w_main [9][C/K][64][K] (tap = ky*3+kx, K-channel chunk, cout, channel in chunk) and
 *                             w_down [C/K][64][K] with K = 64 when C % 64 == 0, else 32.  The dense canvas is never
 *                             materialised.
 *   SCRATCH CONTRACT: `ws` must be ZERO-FILLED before its first use and must not be written by anyone else.  It holds two
 *   (rows, flags) halves used by alternating calls; the consumer of a scatter zeroes, as side work of its own launch, the
 *   rows the PREVIOUS scatter tagged in the other half, so no memset is ever launched.
 * -----------------------------------------------------------------------------------------------*/
/* heal_camera_matrices: the per-camera 3x3 algebra of get_geometry (heter_encoders.py:137-146): fills the `cam_mats`
 *   rows consumed by heal_bev_pool from rots/intrins/post_rots [n,3,3] and trans/post_trans [n,3] (all f32 device),
 *   inverses in closed form (adjugate / determinant) -- one launch, no host round trip.                          */
int heal_camera_matrices(const float* rots, const float* trans, const float* intrins, const float* post_rots,
                         const float* post_trans, int n_cameras, float* cam_mats, void* stream);
size_t heal_bev_pool_workspace(int n_agents, int n_cams, int D, int fH, int fW, int channels,
                               int nx, int ny, int nz);
int heal_bev_pool(const float* depth_logit, const float* feat, const float* frustum,
                  const float* cam_mats, int n_agents, int n_cams, int D, int fH, int fW, int channels,
                  const float* dx_host, const float* bx_host, const int32_t* nx_host,
                  float* out, void* ws, size_t ws_bytes, void* stream);
size_t heal_bev_pool_pm_workspace(int n_agents, int channels, int nx, int ny, int nz);
int heal_bev_pool_scatter(const float* head, int head_stride, const float* frustum, const float* cam_mats, int n_agents,
                          int n_cams, int D, int fH, int fW, int channels, const float* dx_host, const float* bx_host,
                          const int32_t* nx_host, void* ws, size_t ws_bytes, void* stream);

int heal_bev_pool_emit(int n_agents, int channels, const int32_t* nx_host, float* out, void* ws, size_t ws_bytes,
                       void* stream);
int heal_bev_stem_block(int n_agents, int channels, const int32_t* nx_host, const float* w_main, const float* b_main,
                        const float* w_down, const float* b_down, float* out_main, float* out_identity, void* ws,
                        size_t ws_bytes, void* stream);
int heal_bev_pool_pm(const float* head, int head_stride, const float* frustum, const float* cam_mats, int n_agents,
                     int n_cams, int D, int fH, int fW, int channels, const float* dx_host, const float* bx_host,
                     const int32_t* nx_host, float* out, void* ws, size_t ws_bytes, void* stream);

/* Training of the pillar feature net: PFNLayer in training mode (pillar_vfe.py:25-51: Linear(10 -> 64, no bias) ->
 * BatchNorm1d over ALL M x P rows, zeroed padding rows included -> ReLU -> max over the points) and its backward.  max_points <= 32.
 *   heal_pfn_train_blocks(n_voxels)  rows B of the partial-sum outputs below
 *   heal_pfn_moments   partials [B][65]: per-block sums over the rows of the 10 decorated features and of their 55 products
 *                      f_j f_k (j <= k, row-major upper triangle); with z = W f:  mean_c = W_c . s1 / R,  E[z_c^2] = W_c^T S W_c / R
 *   heal_pfn_features  the forward for given scale / shift (batch or running statistics): pillar_feat [M][64], no canvas
 *   heal_pfn_backward  grad_pillar [M][64] -> partials [B][64][12]: per channel A[0..10) = sum dy f_{p*}, B = sum dy,
 *                      Cx = sum dy xhat_{p*} (p* = arg-max row of the forward, dy = grad where y* > 0); the caller forms
 *                      dW, dgamma, dbeta from the block sums (heal_amd/opencood/models/sub_modules/pillar_vfe.py).          */
int heal_pfn_train_blocks(int n_voxels);
int heal_pfn_features(const float* voxels, const int32_t* coords, const int32_t* num_points, int n_voxels, int max_points,
                      const float* weight, const float* bn_scale, const float* bn_shift, float vx, float vy, float vz,
                      float x_offset, float y_offset, float z_offset, float* pillar_feat, void* stream);
int heal_pfn_moments(const float* voxels, const int32_t* coords, const int32_t* num_points, int n_voxels, int max_points,
                     float vx, float vy, float vz, float x_offset, float y_offset, float z_offset, float* partials, void* stream);
int heal_pfn_backward(const float* voxels, const int32_t* coords, const int32_t* num_points, int n_voxels, int max_points,
                      const float* weight, const float* bn_scale, const float* bn_shift, const float* mean, const float* rstd,
                      float vx, float vy, float vz, float x_offset, float y_offset, float z_offset, const float* grad_pillar,
                      float* partials, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K3  SECOND encoder: MeanVFE, sparse 3-D convolution (submanifold and strided), sparse -> dense BEV.
 * Replaces: opencood/models/sub_modules/mean_vfe.py:13-31; the spconv calls of
 *           opencood/models/sub_modules/sparse_backbone_3d.py:11-30,48-91,114-130 (SubMConv3d /
 *           SparseConv3d + BatchNorm1d(eps 1e-3) + ReLU); opencood/models/sub_modules/
 *           height_compression.py:10-26 (.dense() + view [N, C*D, H, W]).
 * A sparse tensor is features [n,C] f32 + indices [n,4] i32 (b,z,y,x) + spatial shape (D,H,W) host.
 * Site sets are kept sorted by linear coordinate.  Shapes / kernel / stride / padding are host int[3]
 * in (z,y,x) order.
 * Row counts: every entry point takes the host value `n` AND an optional device pointer (`n_dev`, `n_out_dev`,
 * `n_in_dev`; NULL = use n).  With a device pointer, `n` is the CAPACITY of the buffers and the live row count is
 * min(*ptr, n): the whole encoder then runs without a host round trip (strided layers produce their site count on the
 * device) and can be captured in a HIP graph.  Rows beyond the live count are unspecified.
 * -----------------------------------------------------------------------------------------------*/
int heal_mean_vfe(const float* voxels, const int32_t* num_points, int n_voxels, int max_points, int n_feat,
                  float* out, const int32_t* n_dev, void* stream);
size_t heal_sp_sort_workspace(int n);
int heal_sp_sort_sites(const int32_t* indices, int n, const int32_t* shape_host, int batch,
                       int32_t* sorted_indices, int32_t* perm, void* ws, size_t ws_bytes, const int32_t* n_dev,
                       void* stream);
/* heal_sp_gather_rows: out[i] = features[perm[i]], rows of `channels` floats: the voxel features re-ordered with the
 *   permutation of heal_sp_sort_sites (spconv keeps features and indices in the caller's order; K3 keeps both sorted by
 *   linear coordinate, sparse_backbone_3d.py:114-116 is where the tensor is formed).  n_dev (may be NULL): live row count. */
int heal_sp_gather_rows(const float* features, const int32_t* perm, int n, int channels, const int32_t* n_dev, float* out,
                        void* stream);
size_t heal_sp_table_capacity(int n);
int heal_sp_hash_build(const int32_t* indices, int n, const int32_t* shape_host, int batch,
                       uint32_t* table_keys, int32_t* table_vals, size_t table_cap, const int32_t* n_dev,
                       void* stream);
/* nbr [n_out, K] i32: input row feeding output site o through tap (kz,ky,kx) row-major, or -1;
 * input coordinate = o*stride - padding + tap (cross-correlation).  Submanifold: pass the input
 * sites as out_indices with stride 1 and padding k/2.                                              */
int heal_sp_neighbors(const int32_t* out_indices, int n_out, const int32_t* ksize_host,
                      const int32_t* stride_host, const int32_t* padding_host, const int32_t* in_shape_host,
                      const int32_t* out_shape_host, int batch, const uint32_t* table_keys,
                      const int32_t* table_vals, size_t table_cap, int32_t* nbr, const int32_t* n_out_dev,
                      void* stream);
/* active output sites of a strided sparse conv (site active iff some active input falls in its
 * receptive field), sorted; out_indices [out_cap,4]; n_out [1] device <- number of active output sites
 * (NOT clamped: a value above out_cap tells the caller that sites were dropped)                     */
size_t heal_sp_out_sites_workspace(int n_in, int kernel_volume);
int heal_sp_out_sites(const int32_t* in_indices, int n_in, const int32_t* ksize_host,
                      const int32_t* stride_host, const int32_t* padding_host, const int32_t* in_shape_host,
                      const int32_t* out_shape_host, int batch, int32_t* out_indices, int out_cap,
                      int32_t* n_out, void* ws, size_t ws_bytes, const int32_t* n_in_dev, void* stream);
/* The same two steps through a RANK STRUCTURE (occupancy bitmap of the grid + prefix counts per 256 cells) instead of hash +
 * sort: for the site sets the encoder itself produces.  heal_sp_out_sites_rank marks the output cells, counts, and emits the
 * sites already sorted; `rank` (heal_sp_rank_bytes(out_shape, batch) bytes, 256-B aligned, caller-owned) then holds the rank
 * structure of that OUTPUT set and answers heal_sp_neighbors_rank queries of every later layer that reads it
 * (row = base[key / 256] + popcount below key).  Same outputs as heal_sp_out_sites / heal_sp_neighbors, bit for bit. */
size_t heal_sp_rank_bytes(const int32_t* shape_host, int batch);
int heal_sp_out_sites_rank(const int32_t* in_indices, int n_in, const int32_t* ksize_host,
                           const int32_t* stride_host, const int32_t* padding_host, const int32_t* in_shape_host,
                           const int32_t* out_shape_host, int batch, int32_t* out_indices, int out_cap,
                           int32_t* n_out, void* rank, size_t rank_bytes, const int32_t* n_in_dev,
                           int32_t* overflow /* optional sticky max(n_out) when n_out > out_cap */, void* stream);
int heal_sp_neighbors_rank(const int32_t* out_indices, int n_out, const int32_t* ksize_host,
                           const int32_t* stride_host, const int32_t* padding_host, const int32_t* in_shape_host,
                           const int32_t* out_shape_host, int batch, const void* rank, size_t rank_bytes,
                           int n_in, const int32_t* n_in_dev, int32_t* nbr, const int32_t* n_out_dev, void* stream);
/* The ROOT site set (the voxels) through a two-level rank structure (round 4): heal_sp_root_rank replaces heal_sp_sort_sites +
 * heal_sp_hash_build -- sorted_indices / perm as heal_sp_sort_sites produces them (sites are unique), by ONE scatter to
 * row = rank(linear coordinate), no sort -- and leaves the structure in `rank` (heal_sp_root_rank_bytes(shape, batch) bytes,
 * 256-B aligned, caller-owned, contents on entry irrelevant: only a 1-bit-per-256-cells directory is cleared per call);
 * heal_sp_neighbors_root is heal_sp_neighbors_rank against that structure.  Replaces spconv's indice-pair generation for the first
 * SubMConv3d / SparseConv3d of sparse_backbone_3d.py:114-118 (`ops.get_indice_pairs` on the voxel coordinates). */
size_t heal_sp_root_rank_bytes(const int32_t* shape_host, int batch);
int heal_sp_root_rank(const int32_t* indices, int n, const int32_t* shape_host, int batch, int32_t* sorted_indices,
                      int32_t* perm, const float* features /* [n, channels] or NULL */, int channels,
                      float* sorted_features /* [n, channels]: features[perm[r]], = heal_sp_gather_rows */, void* rank,
                      size_t rank_bytes, const int32_t* n_dev, void* stream);
int heal_sp_neighbors_root(const int32_t* out_indices, int n_out, const int32_t* ksize_host,
                           const int32_t* stride_host, const int32_t* padding_host, const int32_t* in_shape_host,
                           const int32_t* out_shape_host, int batch, const void* rank, size_t rank_bytes,
                           int n_in, const int32_t* n_in_dev, int32_t* nbr, const int32_t* n_out_dev, void* stream);
/* The rulebook as PAIR TILES (round 6), for the layers with c_in <= 16 (sparse_backbone_3d.py:48-62: conv_input, conv1, the first
 * SparseConv3d of conv2) and the 32 -> 32 submanifold layers of conv2 (heal_sp_conv_tiles_supported): instead of the [n_out][27] table, every `slot_sites` (64; 128 only in a HEAL_BUILD_EXPERIMENTAL=1 library) consecutive output sites own one
 * fixed-stride slot of 64 + 27 slot_sites + 128 words -- word 0: tile count T (a multiple of 4); bytes 4 .. 4+T: the tap of tile i;
 * from word 64: T tiles of 16 pair words, taps ascending, pairs in site order (+ 4 uncounted all-padding tiles).  Pair word = input row << (7 | 8) | site - site0;
 * a padding pair is input row 0 with site = slot_sites.  Only the used prefix of a slot is written and read (6.2 of 27 taps are
 * live at full resolution).  3 x 3 x 3 kernels, rank-structure path (`root` selects the two-level structure of
 * heal_sp_root_rank), n_in < 2^24.  heal_sp_conv_tiles = heal_sp_conv on such a rulebook (fp32 MFMA, fixed summation order: taps
 * ascending per site); heal_sp_tiles_to_neighbors decodes the slots into the [n_out][27] table heal_sp_neighbors_rank / _root
 * would have produced, bit for bit (tests). */
size_t heal_sp_pair_tiles_words(int n_out, int slot_sites);
int heal_sp_neighbor_tiles(const int32_t* out_indices, int n_out, const int32_t* ksize_host, const int32_t* stride_host,
                           const int32_t* padding_host, const int32_t* in_shape_host, const int32_t* out_shape_host,
                           int batch, const void* rank, size_t rank_bytes, int root, int n_in, const int32_t* n_in_dev,
                           int slot_sites, uint32_t* tiles, const int32_t* n_out_dev, void* stream);
int heal_sp_tiles_to_neighbors(const uint32_t* tiles, int n_out, int slot_sites, const int32_t* n_out_dev, int32_t* nbr,
                               void* stream);
int heal_sp_conv_tiles_supported(int c_in, int c_out);
int heal_sp_conv_tiles(const float* feat_in, const uint32_t* tiles, int n_out, int slot_sites, int c_in, int c_out,
                       const float* weight_frag, const float* bn_scale, const float* bn_shift, int relu, float* feat_out,
                       const int32_t* n_out_dev, void* stream);
/* Training: nbr_t [n_in, K] i32 <- the transposed rulebook (nbr_t[i][tap] = o where nbr[o][tap] = i, else -1).  The gradient of
 * heal_sp_conv with respect to its input features is heal_sp_conv itself on (grad_out, nbr_t, weight[tap]^T, scale 1, shift 0,
 * no ReLU): a sparse backward -- no dense grid anywhere.                                                    */
int heal_sp_transpose_neighbors(const int32_t* nbr, int n_out, int kernel_volume, int n_in, int32_t* nbr_t,
                                const int32_t* n_out_dev, void* stream);
/* heal_sp_wgrad: weight gradient of a sparse convolution (spconv's indice_conv_backward, filter part):
 *   dW[tap][ci][co] = sum over the pairs (i = nbr[o][tap] >= 0, o) of feat_in[i][ci] * grad_out[o][co].
 *   partials [heal_sp_wgrad_chunks(n_out)][K][Cin][Cout]: one partial sum per 2048 output rows; the caller adds them (fixed
 *   order: deterministic).  Cin <= 64, Cout in {16, 32, 48, 64}.                                                            */
int heal_sp_wgrad_chunks(int n_out);
int heal_sp_wgrad(const float* feat_in, const float* grad_out, const int32_t* nbr, int n_out, int kernel_volume, int c_in,
                  int c_out, const int32_t* n_out_dev, float* partials, void* stream);
/* feat_out[o] = act(bn_scale * sum_tap W[tap]^T feat_in[nbr[o][tap]] + bn_shift); weight [K,Cin,Cout]
 * (spconv 1.2.1 layout [kz,ky,kx,Cin,Cout]); fp32 MFMA, fixed summation order (bit-reproducible).
 * weight_frag: the same weights re-laid once by heal_sp_weight_fragments (same element count) -- selects the
 * pair-compacted kernel (only real (tap, site) pairs reach the matrix cores); NULL = round-2 kernel on `weight`.
 * Either pointer may be NULL, not both.  n_in rows must be < 2^24.                                   */
int heal_sp_weight_fragments(const float* weight, int kernel_volume, int c_in, int c_out, float* out, void* stream);
int heal_sp_conv(const float* feat_in, const int32_t* nbr, int n_out, int kernel_volume, int c_in, int c_out,
                 const float* weight, const float* weight_frag, const float* bn_scale, const float* bn_shift,
                 int relu, float* feat_out, const int32_t* n_out_dev, void* stream);
size_t heal_sp_to_bev_workspace(int batch, int D, int H, int W);
int heal_sp_to_bev(const float* features, const int32_t* indices, int n, int channels,
                   const int32_t* shape_host, int batch, float* out, void* ws, size_t ws_bytes,
                   const int32_t* n_dev, void* stream);

/* heal_bev_pool_backward: gradient of heal_bev_pool / heal_bev_pool_pm with respect to the depth logits and the image
 *   features (training; the autograd of get_geometry + softmax + voxel_pooling, heter_encoders.py:125-217, without materialising
 *   the [B,N,D,fH,fW,C] lifted tensor).  grad_cells [n_agents*nz*ny*nx, C]: the output gradient re-laid CELL-MAJOR (row =
 *   b*cells + (iz*ny + iy)*nx + ix, the forward's cell key; channel c of row <-> output channel iz*C + c); depth_logit / feat /
 *   frustum / cam_mats / dx / bx / nx as in heal_bev_pool (D <= 64, C <= 256).  -> grad_logit [n_agents*n_cams,D,fH,fW],
 *   grad_feat [n_agents*n_cams,C,fH,fW], every element written.  Cells come from the same fp32 arithmetic as the forward.  */
int heal_bev_pool_backward(const float* grad_cells, const float* depth_logit, const float* feat, const float* frustum,
                           const float* cam_mats, int n_agents, int n_cams, int D, int fH, int fW, int channels,
                           const float* dx_host, const float* bx_host, const int32_t* nx_host, float* grad_logit,
                           float* grad_feat, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K6  per-pixel multi-head attention over agents.
 * Replaces: opencood/models/sub_modules/hmsa.py:110-151 (HGTCavAttention attention core, relation
 *           matrices folded into the projections by the caller) and
 *           opencood/models/fuse_modules/fusion_in_one.py:14-45,126-151 (AttFusion, heads = 1, q=k=v).
 *   q,k,v [n_pix, n_agents, 256] f32 (pixel-major); key_mask [n_agents] i32 DEVICE or NULL (0 = padded
 *   agent, masked to -inf as a key); out [n_pix, out_rows, 256]: rows 0..out_rows-1 of the result
 *   (out_rows = 1 keeps only the ego row, out_rows = n_agents keeps all).
 * -----------------------------------------------------------------------------------------------*/
int heal_agent_attention(const float* q, const float* k, const float* v, const int32_t* key_mask, int n_pix,
                         int n_agents, int channels, int heads, float scale, int out_rows, float* out,
                         int agent_major /* 1: q, k, v, out are [n_agents, n_pix, 256] (the token order of the transformer) */,
                         void* stream);
/* Backward of the same operator (training; the reference differentiates its einsum / bmm formulation through autograd:
 * hmsa.py:131-146, fusion_in_one.py:37-44): grad_out [n_pix, out_rows, 256] (rows >= out_rows carry no gradient) ->
 * grad_q, grad_k, grad_v [n_pix, n_agents, 256], every element written.  The probabilities are recomputed from q and k.       */
int heal_agent_attention_backward(const float* q, const float* k, const float* v, const int32_t* key_mask,
                                  const float* grad_out, int n_pix, int n_agents, int channels, int heads, float scale,
                                  int out_rows, float* grad_q, float* grad_k, float* grad_v, int agent_major, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K7 helpers for the dense BEV stacks.
 * heal_grouped_conv3x3 replaces the 32-group 3x3 convolutions of the ResNeXt bottlenecks
 *   (opencood/models/sub_modules/resblock.py:90-98,110-112: conv2 + bn2 + relu, BatchNorm folded into
 *   weight/bias by the caller); x [n,C,H,W], weight [C, C/groups, 3, 3], bias [C] or NULL, padding 1,
 *   stride 1|2 -> y [n,C,Ho,Wo].
 * heal_bias_act: x = act(x + bias[c] + residual) in place (resblock.py:57-62,113-120: bn (folded),
 *   `out += identity`, relu); residual may be NULL, bias may be NULL.
 * -----------------------------------------------------------------------------------------------*/
int heal_grouped_conv3x3(const float* x, const float* weight, const float* bias, int n, int channels, int groups,
                         int H, int W, int stride, int relu, float* y, void* stream);
int heal_bias_act(float* x, const float* bias, const float* residual, int n, int channels, int HW, int relu,
                  void* stream);

/* heal_upsample2x_bilinear: nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) of the
 *   Lift-Splat `Up` block (opencood/models/sub_modules/lss_submodule.py:21-22,33); x [n,C,H,W] -> y [n,C,2H,2W] */
int heal_upsample2x_bilinear(const float* x, int n, int channels, int H, int W, float* y, void* stream);

/* heal_depthwise_conv: depthwise k x k (k = 3|5 at stride 1|2, k = 7 at stride 1: the ConvNeXt aligner's dwconv,
 *   feature_alignnet_modules.py:314) with explicit top/left zero padding, bias and
 *   activation (0 none, 1 ReLU, 2 SiLU): the MBConv depthwise stage of the EfficientNet-b0 trunk
 *   (lss_submodule.py:93-105); x [n,C,H,W], weight [C,1,k,k] -> y [n,C,Ho,Wo].
 *   channel_sums (NULL or [n,C,T] f32, T = ceil(Wo/32) * ceil(Ho/8) output tiles): every block stores the sum of its tile's
 *   activated outputs -- the squeeze of the squeeze-excite stage that follows in an MBConv block, folded into this launch
 *   (plain stores, every word written); heal_se_gate(scale = 1/(Ho*Wo), tiles = T) adds the tiles up in a fixed order.  */
int heal_depthwise_conv(const float* x, const float* weight, const float* bias, int n, int channels, int H, int W,
                        int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, float* y,
                        float* channel_sums, void* stream);

/* heal_layernorm_nchw: LayerNorm over the channel axis of an NCHW map, y = (x - mean_c) / sqrt(var_c + eps) * gamma + beta
 *   (biased variance) -- the `norm` of the ConvNeXt aligner block (feature_alignnet_modules.py:12-31,318-321), which the
 *   reference evaluates as permute -> F.layer_norm -> ... -> permute.  x,y [n,C,H,W].                               */
int heal_layernorm_nchw(const float* x, const float* gamma, const float* beta, int n, int channels, int HW, float eps,
                        float* y, void* stream);

/* heal_channel_dot: pointwise convolution to ONE output channel, y[n][p] = bias[0] + sum_c weight[c] x[n][c][p] -- the occupancy
 *   heads `single_head_i` of PyramidFusion (opencood/models/fuse_modules/pyramid_fuse.py:89-91, nn.Conv2d(C, 1, 1)) whose logits
 *   feed heal_warp_fuse.  x [n,C,H,W] f32, y [n,1,H,W], HW = H*W % 4 == 0, 16-B aligned; bias may be NULL.                 */
int heal_channel_dot(const float* x, const float* weight, const float* bias, int n, int channels, int HW, float* y,
                     void* stream);

/* heal_se_gate: squeeze-excite gate of the EfficientNet MBConv block (efficientnet_pytorch MBConvBlock as called from
 *   lss_submodule.py:93-105): gate [n,C] = sigmoid(W_expand silu(W_reduce mean + b_reduce) + b_expand) from the spatial
 *   squeezed input = scale * sum_t mean[n][c][t]; W_reduce [S,C]; w_expand_t = W_expand^T laid out [S,C] (coalesced
 *   columns); S <= 64.  tiles = 1, scale = 1: `mean` is the spatial mean [n,C]; tiles = T, scale = 1/(Ho*Wo): `mean` holds the
 *   per-tile sums [n,C,T] heal_depthwise_conv leaves.  Feeds heal_conv1x1's in_scale.                                 */
int heal_se_gate(const float* mean, const float* w_reduce, const float* b_reduce, const float* w_expand_t,
                 const float* b_expand, int n, int channels, int squeezed, float scale, int tiles, float* gate,
                 void* stream);

/* heal_conv1x1: pointwise convolution with fused prologue/epilogue,
 *     y = act(W (in_scale . x) + bias (+ residual)),
 *   i.e. the conv1x1 + BatchNorm (+ identity) + ReLU sequences of the ResNeXt bottleneck
 *   (opencood/models/sub_modules/resblock.py:95-121) and the expand / squeeze-excite-scale + project (+ skip) stages of
 *   the EfficientNet MBConv blocks of the Lift-Splat trunk (lss_submodule.py:93-105), BatchNorm folded into W / bias by
 *   the caller.  x [n,Cin,H,W] f32 NCHW; stride 1 | 2 (the 1x1 stride-2 `downsample` of resblock.py:160-165: every
 *   second pixel); y/residual [n,Cout,Ho,Wo], Ho = (H-1)/stride+1; Ho*Wo % 4 == 0 (stride 1) or Wo % 4 == 0 (stride 2);
 *   in_scale [n,Cin] or NULL (per-image, per-input-channel gate); bias [Cout] or NULL; act 0 none | 1 ReLU | 2 SiLU | 3 GELU (erf).
 *   weight_frag = W zero-padded to [Mpad = ceil64(Cout), Kpad = ceil32(Cin)] in MFMA A-fragment order
 *   with four k-steps of a lane contiguous (ABI 3): frag[mt][ks / 4][lane][ks % 4] = W[mt*16 + (lane & 15)][ks*4 + (lane >> 4)],
 *   mt < Mpad/16, ks < Kpad/4 (heal_amd.ops.conv1x1_fragments).
 *   out_pixel_major != 0: y is written as [n, Ho*Wo, Cout] (a pixel's channels contiguous; Cout % 4 == 0, no residual) --
 *   the layout heal_bev_pool_pm reads, produced by the fused image_head | depth_head convolution of CamEncode
 *   (lss_submodule.py:113-131) so that the lift never needs a transposition pass.                              */
int heal_conv1x1(const float* x, const float* weight_frag, const float* bias, const float* residual,
                 const float* in_scale, int n, int cin, int cout, int H, int W, int stride, int act,
                 int out_pixel_major, float* y, void* stream);

/* heal_conv3x3_same: heal_conv3x3 with the zero padding given as (pad_t, pad_l) rows / columns IN FRONT of the map (0 | 1) and
 *   whatever the output size needs behind it (at most one): TensorFlow-style "same" padding, which for a stride-2 convolution on
 *   an even map pads only behind -- the 3x3 / stride-2 stem of the EfficientNet-b0 camera trunk (efficientnet_pytorch
 *   Conv2dStaticSamePadding as used by lss_submodule.py:58-60) -- with bias and activation (0 none | 1 ReLU | 2 SiLU) fused.
 *   Same weight fragment layout as heal_conv3x3.                                                                        */
int heal_conv3x3_same(const float* x, const float* weight_frag, const float* bias, int n, int cin, int cout, int H, int W,
                      int stride, int pad_t, int pad_l, int Ho, int Wo, int act, float* y, void* stream);

/* heal_conv1x1_splitk: heal_conv1x1 (stride 1, NCHW) for SMALL maps with a DEEP reduction -- the MBConv projections of the
 *   EfficientNet trunk at 1/16 .. 1/32 resolution (1152 -> 192 at 12 x 16 x 4 pixels is 36 blocks of 36 K-chunks each) -- with
 *   the K chunks split over `ksplit` blocks: partial sums go to the workspace [ksplit][n][Cout][HW], a second launch adds them
 *   in split order (deterministic) and applies bias / residual / activation.  ksplit in [2, ceil32(Cin)/32] with no empty
 *   split; workspace from heal_conv1x1_splitk_workspace, 16-B aligned; H*W % 4 == 0.                                  */
/* heal_stem7x7: the ResNet image stem in one kernel -- Conv2d(cin <= 4 -> 64, kernel 7, stride 2, padding 3) + bias (folded
 *   BatchNorm) + ReLU and, with pool != 0, MaxPool2d(3, stride 2, padding 1) (lss_submodule.py:153-161,196-210; torchvision
 *   resnet101 conv1 / bn1 / relu / maxpool).  x: images `image_stride` floats apart, the FIRST cin channels of each are read in
 *   place ([n, >= cin, H, W]: no contiguous copy of a channel slice).  weight_frag [4][ceil(cin*49/4)][64]: the [64, cin*49]
 *   weight matrix in 16x16x4 A-fragment order, frag[mt][ks][16 lk + ln] = W[16 mt + ln][4 ks + lk] (zero beyond cin*49).
 *   y: [n, 64, Hp, Wp] with Hp = ((H - 1) / 2) / 2 + 1 ... (pool) or [n, 64, (H - 1) / 2 + 1, (W - 1) / 2 + 1] (no pool).          */
int heal_stem7x7(const float* x, long long image_stride, int n, int cin, int H, int W, const float* weight_frag,
                 const float* bias, int pool, float* y, void* stream);
/* heal_conv1x1_split (round 6, OPT-IN prototype: HEAL_ARITH=bf16x6 | bf16x9 in the host mirror; never the default): the same pointwise
 *   convolution with fp32 inputs / outputs and fp32 accumulation, evaluated on the BF16 matrix cores by splitting both operands into three
 *   bf16 numbers (a = a_h + a_m + a_l exactly) and summing n_products = 6 (terms down to 2^-16 of the product) or 9 (all: exact products)
 *   partial products.  weight_frag: the three bf16 planes of W [Cout, Cin] in fragment order (heal_amd.ops.conv1x1_split_fragments:
 *   [Cout/128][Cin/32][2][4][3][64 lanes][8 bf16]); stride 1, NCHW in and out; cin % 32 == 0, cout % 128 == 0; act 0 none | 1 ReLU |
 *   2 SiLU | 3 GELU (erf).  Error against fp64 and the exact-fp32 kernel: tests/test_gpu_kernels.py::test_conv1x1_split_*.          */
int heal_conv1x1_split_supported(int cin, int cout, int H, int W);
int heal_conv1x1_split(const float* x, const void* weight_frag, const float* bias, const float* residual, int n, int cin, int cout,
                       int H, int W, int act, int n_products, float* y, void* stream);
size_t heal_conv1x1_splitk_workspace(int n, int cout, int H, int W, int ksplit);
int heal_conv1x1_splitk(const float* x, const float* weight_frag, const float* bias, const float* residual,
                        const float* in_scale, int n, int cin, int cout, int H, int W, int act, int ksplit, float* y, void* ws,
                        size_t ws_bytes, void* stream);

/* heal_conv1x1_d2s: heal_conv1x1 (stride 1, no residual / gate) whose epilogue writes DEPTH-TO-SPACE INTO A CHANNEL SLICE of a
 *   wider NCHW tensor: output channel co of pixel (h, w) goes to channel dst_channel_offset + co / k^2, pixel
 *   (h k + (co % k^2) / k, w k + co % k) of y [n, dst_channels, H k, W k].  This is the deblock of the BEV backbones --
 *   ConvTranspose2d(kernel = stride = k) + BatchNorm + ReLU (base_bev_backbone_resnet.py:49-74,128-131; folded weight laid out
 *   [Cout k^2, Cin] by the caller) -- written straight into the torch.cat of the upsampled levels, so neither the pixel
 *   shuffle nor the concatenation is a separate pass.  k = 1: a plain channel-offset write.  W % 4 == 0.            */
int heal_conv1x1_d2s(const float* x, const float* weight_frag, const float* bias, int n, int cin, int cout, int H, int W,
                     int act, int k, int dst_channels, int dst_channel_offset, float* y, void* stream);

/* heal_conv3x3: dense 3x3 convolution, padding 1, stride 1 | 2, with the epilogue fused:
 *     y = act(W * x + bias (+ residual)),
 *   i.e. the conv3x3 + BatchNorm (+ identity) + ReLU sequences of BasicBlock (opencood/models/sub_modules/resblock.py:18-64),
 *   the conv + ReLU pairs of DoubleConv (downsample_conv.py:7-27: the 384->256 / 256->256 shrink header at 256x256), the
 *   plain Conv-BN-ReLU stacks of BaseBEVBackbone (base_bev_backbone.py:6-124), `Up` of the Lift-Splat camera encoder
 *   (lss_submodule.py:17-36) and the 3x3 convolutions of the ResNet101 stem, BatchNorm folded into W / bias by the caller.
 *   fp32 in, fp32 accumulate on v_mfma_f32_16x16x4_f32 (implicit GEMM, no reduced precision).
 *   x [n,Cin,H,W] f32 NCHW; y / residual [n,Cout,Ho,Wo] with Ho = (H-1)/stride + 1; bias [Cout] or NULL; relu 0 | 1.
 *   weight_frag = W [Cout,Cin,3,3] zero-padded to [Mpad = ceil64(Cout), Kpad = ceil8(Cin)] in MFMA A-fragment order
 *   frag[mb][chunk][tap][ks][mt][lane] = W[mb*64 + mt*16 + (lane & 15)][chunk*8 + ks*4 + (lane >> 4)][tap],
 *   mb < Mpad/64, chunk < Kpad/8, tap = 3*ky + kx, ks < 2, mt < 4, lane < 64 (16-B aligned).                        */
int heal_conv3x3(const float* x, const float* weight_frag, const float* bias, const float* residual, int n, int cin,
                 int cout, int H, int W, int stride, int relu, float* y, void* stream);

/* heal_conv_gemm: the same dense convolutions (3x3 padding 1, or 1x1; stride 1 | 2; y = act(W * x + bias (+ residual))) as an
 *   implicit GEMM on 128 x 128 x 32 tiles of v_mfma_f32_32x32x2_f32 -- the formulation for the LARGE stride-2 layers
 *   (BaseBEVBackbone stage heads, base_bev_backbone.py:49-74; the 384 -> 256 shrink_header of HeterModelBaseline,
 *   downsample_conv.py:7-49), where the long reduction (9 Cin) amortises its 128 x 128 tiles.  weight_tap_major = W re-laid
 *   [Cout, k*k, Cin] (tap = k ky + kx); Cout % 128 == 0, Cin % 32 == 0, output width % 4 == 0.  Round 6: ksize 7 with stride 2
 *   (padding 3) as well -- BevEncode's stem Conv2d(inC, 64, 7, 2, 3) of the old-style Lift-Splat model (lss_submodule.py:242), the last
 *   convolution any mirrored model left to the library (the caller pads its 64 output channels to 128 zero rows).                    */
int heal_conv_gemm(const float* x, const float* weight_tap_major, const float* bias, const float* residual, int n, int cin,
                   int cout, int H, int W, int ksize, int stride, int relu, float* y, void* stream);

/* heal_grouped16_conv3x3: the 32-group 3x3 convolution of the ResNeXt bottlenecks (resblock.py:90-98,110-112; stride 1,
 *   padding 1, folded BatchNorm bias, ReLU) on the matrix cores, for 16 or 8 channels per group: channels are processed in
 *   16-channel super-groups (one group of 16, or two groups of 8 with block-diagonal weights).  x, y [n,C,H,W].
 *   weight_frag[sg][tap][ks][lane] = Wsg[co = lane & 15][ci = ks*4 + (lane >> 4)][tap], sg < C/16, where Wsg is the
 *   16x16x9 weight of the super-group (zero outside the diagonal blocks when a group has 8 channels); 16-B aligned.   */
int heal_grouped16_conv3x3(const float* x, const float* weight_frag, const float* bias, int n, int channels, int H, int W,
                           int relu, float* y, void* stream);

/* heal_grouped_small_conv3x3: the same 32-group 3x3 convolution (stride 1 | 2, padding 1, folded BatchNorm bias, ReLU) for 4 or 8
 *   channels per group -- the 128- and 256-wide ResNeXt stages of PyramidFusion (resblock.py:90-98,110-112) -- on the 16-block
 *   v_mfma_f32_4x4x1_16b_f32: a block is 4 output channels x 4 pixels x 1 input channel, so no multiply is spent on the zeros
 *   of a block-diagonal weight.  x [n,C,H,W], y [n,C,Ho,Wo] (Ho = (H-1)/stride + 1), C % 16 == 0, W % 4 == 0, Wo % 4 == 0,
 *   x and y 16-B aligned.
 *   weight_q[sg][tap][ci][co] = W[sg*16 + co][ci][tap], sg < C/16, tap = 3*ky + kx, ci < group_channels, co < 16.      */
int heal_grouped_small_conv3x3(const float* x, const float* weight_q, const float* bias, int n, int channels,
                               int group_channels, int H, int W, int stride, int relu, float* y, void* stream);

/* heal_conv3x3_winograd: the same operator for stride 1 evaluated with the Winograd F(2x2,3x3) minimal-filtering transform
 *   on the matrix cores (16 transform-domain GEMMs, 2.25x fewer MFMAs than the implicit GEMM of heal_conv3x3; fp32
 *   throughout, results differ from the direct evaluation by rounding only, ~1e-6 relative).  Same tensors as heal_conv3x3.
 *   u_frag = U = G g G^T of every (co, ci) filter (G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]), zero-padded to
 *   [Mpad = ceil64(Cout), Kpad = ceil8(Cin)], lane-major per wave; `waves` = 8 (block = 64 channels x 16x16 pixels, two
 *   transform positions per wave) or 4 (64 channels x 8x16 pixels, four positions per wave, two blocks per CU); XW = 16/waves:
 *   frag[mb][chunk][w][lane][(xi_i*2 + ks)*4 + mt] = U[mb*64 + mt*16 + (lane & 15)][chunk*8 + ks*4 + (lane >> 4)][xi = XW*w + xi_i],
 *   w < waves, xi_i < XW, xi = 4a + b indexes the 4x4 transform domain (16-B aligned).  The fragment order depends on `waves`. */
int heal_conv3x3_winograd(const float* x, const float* u_frag, const float* bias, const float* residual, int n, int cin,
                          int cout, int H, int W, int relu, int waves, float* y, void* stream);
/* heal_conv3x3_winograd_kc: the same with `kc` input channels per chunk of the K loop: 8 (= heal_conv3x3_winograd) or 16 (round 5: half
 *   the barriers and operand waits per MFMA, 112 KB of LDS; waves = 8 and cin %% 16 == 0 only; measured 4-13 % SLOWER, so the kc = 16
 *   instantiation exists only in a HEAL_BUILD_EXPERIMENTAL=1 library -- the shipped one rejects kc = 16).  Fragment order with KS = kc / 4:
 *   frag[mb][chunk][w][lane][(xi_i*KS + ks)*4 + mt] = U[mb*64 + mt*16 + (lane & 15)][chunk*kc + ks*4 + (lane >> 4)][xi = XW*w + xi_i]. */
int heal_conv3x3_winograd_kc(const float* x, const float* u_frag, const float* bias, const float* residual, int n, int cin,
                             int cout, int H, int W, int relu, int waves, int kc, float* y, void* stream);
/* heal_conv3x3_winograd_splitk (round 6): heal_conv3x3_winograd for SMALL maps with a DEEP reduction -- the 432 -> 512 and 512 -> 512
 *   layers of the camera trunk's Up block (lss_submodule.py:33-50) at 4 x 24 x 32 pixels are 192 blocks of 54 / 64 K-chunks each --
 *   with the chunks of 8 input channels split over `ksplit` blocks.  The output transform is linear: every block writes its PARTIAL
 *   output to the workspace [ksplit][n][Cout][H W], a second launch adds the partials in split order (deterministic) and applies
 *   bias / residual / ReLU.  ksplit in [2, ceil(cin / 8)] with no empty split; H*W % 4 == 0; workspace from
 *   heal_conv3x3_winograd_splitk_workspace, 16-B aligned.                                                                */
size_t heal_conv3x3_winograd_splitk_workspace(int n, int cout, int H, int W, int ksplit);
int heal_conv3x3_winograd_splitk(const float* x, const float* u_frag, const float* bias, const float* residual, int n, int cin,
                                 int cout, int H, int W, int relu, int waves, int ksplit, float* y, void* ws, size_t ws_bytes,
                                 void* stream);

/* ---- pcdet rotated-BEV box ops ------------------------------------------------------------
 * Replace opencood/pcdet_utils/iou3d_nms/src/iou3d_nms_kernel.cu:104-234 (box_overlap, iou_bev), :236-265
 * (boxes_overlap_kernel, boxes_iou_bev_kernel), :267-375 (nms_kernel, nms_normal_kernel) and the host mask walk of
 * iou3d_nms.cpp:74-125,128-188, i.e. the `iou3d_nms_cuda` module bound by iou3d_nms_utils.py:32-46,109-181,252-289.
 * Boxes are rows [x, y, z, dx, dy, dz, heading] f32.  pcdet semantics: fp32, corner test inflated by 1e-2,
 * points sorted by polar angle around their centroid.
 *
 * heal_boxes_bev_matrix: out[n,m] = overlap area (mode 0: boxes_overlap_bev_gpu), rotated IoU (mode 1:
 *   boxes_iou_bev_gpu) or axis-aligned IoU (mode 2: iou_normal).
 * heal_nms_bev: greedy NMS over boxes ALREADY sorted by descending score (the caller's `boxes[order]`,
 *   iou3d_nms_utils.py:262-270); rotated != 0 -> nms_gpu, 0 -> nms_normal_gpu.  keep [n] i64 (indices into the
 *   sorted boxes, ascending) and *num_keep stay on the device: no mask copy to the host, no per-call allocation. */
int heal_boxes_bev_matrix(const float* boxes_a, int n, const float* boxes_b, int m, int mode, float* out,
                          void* stream);
size_t heal_nms_bev_workspace(int n);
int heal_nms_bev(const float* boxes_sorted, int n, float thresh, int rotated, void* workspace,
                 size_t workspace_bytes, long long* keep, int* num_keep, void* stream);

/* heal_nms_quads: rotated NMS over projected footprints, the body of opencood/utils/box_utils.py:693-738 (nms_rotated:
 *   greedy, IoU of the quads = corners[0:4,:2] as convex polygons in fp64 like shapely/GEOS, cast to f32, `> thresh`
 *   suppresses).  quads_sorted [n,4,2] f32 ALREADY in descending-score order (the caller's `scores.argsort()[::-1][:1000]`);
 *   keep [n] i64 <- indices into that order, ascending (= pick order); *num_keep device.                          */
size_t heal_nms_quads_workspace(int n);
int heal_nms_quads(const float* quads_sorted, int n, float thresh, void* workspace, size_t workspace_bytes,
                   long long* keep, int* num_keep, void* stream);

/* heal_window_attention: fused window attention of the V2X-ViT pyramid (opencood/models/sub_modules/mswin.py:46-80):
 *   out[l,y,x,h*d:(h+1)*d] = (softmax(scale * Q K^T + pos_bias) V) inside every window x window tile, per agent and head.
 *   qkv [n_agents,H,W,3*heads*dim_head] f32 = the packed to_qkv projection (q | k | v chunks, each (head, dim));
 *   pos_bias [T,T] (T = window^2, gathered relative-position embedding) or NULL; out [n_agents,H,W,heads*dim_head].
 *   (window, dim_head) in {(4,16),(8,32),(16,64),(4,32),(8,16),(8,64),(4,64)}; H, W multiples of window.          */
int heal_window_attention(const float* qkv, const float* pos_bias, int n_agents, int H, int W, int heads,
                          int dim_head, int window, float scale, float* out, void* stream);
/* Backward of heal_window_attention (training; the reference differentiates mswin.py:64-78 through autograd).  OPT-IN in the host
 * mirror until measured (HEAL_WATTN_GRAD=kernel).  out = the forward's result, grad_out its gradient, both [L,H,W,heads*dim_head];
 * grad_qkv [L,H,W,3*heads*dim_head] is written completely; grad_bias [T,T] (or NULL) is ACCUMULATED with atomics into a buffer the
 * caller zeroes (the sum over agents, windows and heads).  Workspace: (max, 1/sum, dO.O) per query and head.                      */
size_t heal_window_attention_backward_workspace(int n_agents, int H, int W, int heads);
int heal_window_attention_backward(const float* qkv, const float* pos_bias, const float* out, const float* grad_out, int n_agents,
                                   int H, int W, int heads, int dim_head, int window, float scale, float* grad_qkv,
                                   float* grad_bias, void* ws, size_t ws_bytes, void* stream);

/* ---- V2X-ViT linear algebra (opencood/models/sub_modules/base_transformer.py:7-40, hmsa.py:38-151, mswin.py:46-122,
 * split_attn.py:6-62, v2xvit_basic.py:158-192): token-major fp32 GEMM on v_mfma_f32_32x32x2_f32 with the LayerNorm of PreNorm
 * in the prologue and bias / GELU / residual / re-layout in the epilogue, instead of library GEMMs with ATen kernels between.
 * heal_ln_stats: stats [n_tokens, 2] <- (mean, 1 / sqrt(var + eps)) of every token of x [n_tokens, channels].
 * heal_linear: out = act(norm(x) W^T * colscale + bias) + residual.
 *   x [n_tokens, n_in] rows lda floats apart (x_part_cols > 0: channel block k / x_part_cols of a token is read from
 *   x + block * x_part_stride, i.e. the reduction runs over several [n_tokens, x_part_cols] tensors); ln_stats (or NULL): x is replaced by (x - mean) * rstd per token (fold the
 *   LayerNorm's gamma into W and beta W^T into bias); weight [n_out, n_in] (nn.Linear layout); bias [n_out], or
 *   [groups, n_out] with bias_per_group; colscale (or NULL) [groups, n_in / colscale_part, n_out]: weight[n][k] is multiplied by
 *   colscale[group][k / colscale_part][n], group = token / group_rows (the split-attention merge: three to_out projections and
 *   the per-agent softmax weights as one K = 3 C GEMM); residual (or NULL) rows ldr apart, indexed like the output; act 0 none,
 *   1 GELU (erf), 2 ReLU (applied before the residual).  Output addressing: token t goes to row (t % map_inner) * map_outer +
 *   t / map_inner (map_inner = 0: row t), column c to out + (c / part_cols) * part_stride + row * ldo + c % part_cols.
 *   n_out multiple of 128, n_in multiple of 32, group_rows multiple of 128.
 * heal_split_attn_weights: branches = the three window-attention outputs [3][groups * rows_per_group, channels] (part_stride
 *   floats apart), BEFORE their to_out projections w_out [3, C, C] / b_out [3, C]; -> scale [groups, 3, C] (softmax over the
 *   three branches of fc2(relu(LN(fc1(global average of the projected sum))))) and bias [groups, C] = sum scale * b_out.     */
int heal_ln_stats(const float* x, int n_tokens, int channels, float eps, float* stats, void* stream);
int heal_linear(const float* x, int lda, int x_part_cols, long long x_part_stride, const float* ln_stats,
                const float* weight, const float* bias,
                int bias_per_group, const float* colscale, int colscale_part, int group_rows, const float* residual,
                int ldr, float* out, int ldo, int n_tokens, int n_out, int n_in, int map_inner, int map_outer,
                int part_cols, long long part_stride, int act, void* stream);
size_t heal_split_attn_workspace(int groups, int rows_per_group, int channels);
int heal_split_attn_weights(const float* branches, long long part_stride, int groups, int rows_per_group, int channels,
                            const float* w_out, const float* b_out, const float* fc1, const float* ln_gamma,
                            const float* ln_beta, float eps, const float* fc2, float* colsum_ws, float* scale,
                            float* bias, void* stream);
/* The same operator in its two halves, for a scene whose tokens are spread over ranks as row stripes (heal_amd/dist.py,
 * ShardedBaselineStriped; the only step of the V2X-ViT encoder that looks beyond a window, mswin.py:118-122 -> split_attn.py:43-62):
 * heal_split_attn_colsum: colsum [groups, 3, ceil(rows_per_group / 512), channels] <- per-chunk column sums of the LOCAL tokens;
 * the ranks all-gather these (12 KB per group and rank), and heal_split_attn_weights_from_colsum reduces colsum [n_parts, groups,
 * 3, chunks, channels] part-major, chunk by chunk -- the order of the unsharded call when rows_per_part is a multiple of 512 --
 * over n_parts * rows_per_part tokens per group.                                                                               */
int heal_split_attn_colsum(const float* branches, long long part_stride, int groups, int rows_per_group, int channels,
                           float* colsum, void* stream);
int heal_split_attn_weights_from_colsum(const float* colsum, int n_parts, int groups, int rows_per_part, int channels,
                                        const float* w_out, const float* b_out, const float* fc1, const float* ln_gamma,
                                        const float* ln_beta, float eps, const float* fc2, float* scale, float* bias,
                                        void* stream);

/* ---- training-side anchor labelling --------------------------------------------------------
 * heal_label_assign: the IoU / assignment core of VoxelPostprocessor.generate_label
 *   (opencood/data_utils/post_processor/voxel_postprocessor.py:139-165), replacing the Cython bbox_overlaps
 *   (opencood/utils/box_overlaps.pyx:17-57) and the numpy argmax / where / unique bookkeeping around it.
 *   anchor_boxes [n_anchors,4], gt_boxes [n_gt,4] f32: axis-aligned stand-up boxes (x1,y1,x2,y2), n_gt <= 512.
 *   assigned [n_anchors] i32 <- gt index of a positive anchor (first gt with IoU > pos_threshold; else, for the best
 *   anchor of a gt with IoU > 0, the smallest such gt), -1 otherwise; neg [n_anchors] u8 <- 1 where every IoU is
 *   below neg_threshold, except the best anchors.  IoU arithmetic bit-exact with the Cython routine.              */
size_t heal_label_assign_workspace(int n_gt);
int heal_label_assign(const float* anchor_boxes, int n_anchors, const float* gt_boxes, int n_gt,
                      float pos_threshold, float neg_threshold, int32_t* assigned, uint8_t* neg, void* ws,
                      size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HEAL_AMD_H */
