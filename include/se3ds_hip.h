/* se3ds_hip.h -- C ABI of libse3ds_hip.so: the MI355X (gfx950) kernels behind the SE3DS hot
 * path.  The reference (google-research/se3ds) has no FFI of its own: its "native layer" is
 * the set of TensorFlow / tensorflow-addons ops its Python calls.  Each entry point below
 * replaces the TF op sequence at the cited reference lines; INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - raw DEVICE pointers + explicit sizes; `stream` is a hipStream_t passed as void*
 *   - returns 0 on success or a negative SE3DS_E_* code; never throws, never allocates,
 *     never synchronises the stream, owns nothing; re-entrant for distinct streams
 *   - scratch memory comes from the caller (`*_workspace_bytes` query functions)
 *   - tensors are dense, row-major, NHWC for images; fp32 unless a dtype code says otherwise
 */
#ifndef SE3DS_HIP_H_
#define SE3DS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SE3DS_OK 0
#define SE3DS_E_BADSHAPE (-1)
#define SE3DS_E_BADDTYPE (-2)
#define SE3DS_E_WORKSPACE (-3)
#define SE3DS_E_LAUNCH (-4)
#define SE3DS_E_UNSUPPORTED (-5)

/* element type codes */
#define SE3DS_F32 0
#define SE3DS_I32 1
#define SE3DS_U8 2
#define SE3DS_BF16 3

/* Library / build identification: returns a static string "se3ds_hip <abi> gfx950". */
const char* se3ds_version(void);
/* HIP error string of the last failed launch on this thread (or ""). */
const char* se3ds_last_error(void);

/* ======================================================================================
 * Geometry (point-cloud warp)
 * ====================================================================================== */

/* equirectangular_to_pointcloud -- reference utils/pano_utils.py:164-242 (core :219-242).
 * feats (N,H,W,C) of `feat_dtype`, depth (N,H,W) fp32 in [0,1].  sin_el/cos_el (H) and
 * sin_hd/cos_hd (W) are the fp32 sin/cos tables of the half-pixel-centre elevation/heading
 * grids (:211-218), built on the host.  `position` (N,3) is optional (may be NULL) and is
 * ADDED to xyz after the unprojection (models/models.py:225-226 `xyz1 += position`, the
 * 4th row gets += 0).  Outputs: xyz1 (N,4,H*W) fp32, feats_out (N,H*W,C) same dtype with
 * `void_class` where depth is not in (0,1). */
int se3ds_unproject_equirect(const void* feats, int feat_dtype, const float* depth,
                             const float* sin_el, const float* cos_el, const float* sin_hd,
                             const float* cos_hd, const float* position, int n, int height,
                             int width, int channels, float void_class, float depth_scale,
                             float* xyz1, void* feats_out, void* stream);

/* Scratch bytes for the two splat entry points below. */
size_t se3ds_splat_workspace_bytes(int n, int64_t m, int height, int width, int channels);

/* project_feats_to_equirectangular -- reference utils/pano_utils.py:117-161 fused with
 * utils/point_cloud_utils.py:90-183 (project_to_feat) and, optionally, the mask of
 * models/models.py:282-287.
 * xyz1 (N,4,M) fp32 world coords; `offset` (N,3) optional, SUBTRACTED first
 * (models.py:273-275 / eval_metric.py:162 `memory - position`).  feats (N,M,C) of
 * `feat_dtype` (cast to fp32 as pano_utils.py:159 does).  Outputs: depth (N,H,W) in [0,1],
 * feat (N,H,W,C) fp32, mask (N,H,W) fp32 {0,1} or NULL
 * (mask = depth in (0,1) and all(feat != mask_void)).
 * Semantics kept from the reference: invalid / culled points scatter into flat index 0
 * (batch 0, pixel (0,0)); 0.1 m tolerance; per-channel max over all survivors. */
int se3ds_project_equirect(const float* xyz1, const float* offset, const void* feats,
                           int feat_dtype, int n, int64_t m, int channels, int height, int width,
                           float depth_scale, float input_void, float output_void, float* depth,
                           float* feat, float* mask, float mask_void, void* workspace,
                           size_t workspace_bytes, void* stream);

/* project_to_feat -- reference utils/point_cloud_utils.py:90-183 on already transformed
 * coordinates (N,4,M) = (x, y, z, 1).  Same outputs/semantics as above. */
int se3ds_project_to_feat(const float* coords, const void* feats, int feat_dtype, int n, int64_t m,
                          int channels, int height, int width, float depth_scale,
                          float input_void, float output_void, float* depth, float* feat,
                          float* mask, float mask_void, void* workspace, size_t workspace_bytes,
                          void* stream);

/* Debug/parity tap: first-stage flat indices of the last se3ds_project_* call that used
 * `workspace` (int32 per point: index inside the image v*W+u, or -1 = sink) and z. */
int se3ds_splat_debug_indices(const void* workspace, int n, int64_t m, int32_t* idx_out,
                              float* z_out, void* stream);

/* get_filtered_coords_and_feats -- reference utils/point_cloud_utils.py:32-87 (perspective,
 * 90 deg HFOV).  feats (N,H,W,C) int32, depth (N,H,W) fp32; xs (W), ys (H) are the fp32
 * linspace(-1,1) grids and kinv (4x4 row-major) the inverse intrinsics, built on the host.
 * Outputs xyz (N,4,H*W) fp32, feats_out (N,H*W,C) fp32. */
int se3ds_unproject_perspective(const int32_t* feats, const float* depth, const float* xs,
                                const float* ys, const float* kinv, int n, int height, int width,
                                int channels, float depth_scale, float* xyz, float* feats_out,
                                void* stream);

/* tfa.image.interpolate_bilinear -- called at utils/pano_utils.py:339,412,472.
 * grid (B,H,W,C) fp32, query (B,Q,2) fp32, out (B,Q,C) fp32.  indexing_xy: 0 = 'ij'
 * (query = (row, col)), 1 = 'xy' (query = (x, y)). */
int se3ds_interp_bilinear(const float* grid, const float* query, int b, int height, int width,
                          int channels, int64_t q, int indexing_xy, float* out, void* stream);

/* Coordinate generation of rotate_pano -- utils/pano_utils.py:326-338.  rays (3,Q) fp32
 * (equirectangular_pixel_rays), matrix (N,3,3); out (N,Q,2) = (pitch_px, heading_px) for a
 * source pano of src_h x src_w. */
int se3ds_rotate_coords(const float* rays, const float* matrix, int n, int64_t q, int src_h,
                        int src_w, float* out, void* stream);

/* Coordinate generation of project_perspective_image -- utils/pano_utils.py:387-402.
 * rays (3,Q), world_to_image (3,3); out (Q,2) = xy/z where z > 0 else -1, optional
 * round-half-even, plus `add` (1.0 when the image was padded by one pixel, :409). */
int se3ds_perspective_coords(const float* rays, const float* w2i, int64_t q, int round_nearest,
                             float add, float* out, void* stream);

/* Coordinate generation of get_perspective_from_equirectangular_image --
 * utils/pano_utils.py:459-469.  m (3,3) = K^-T . R applied as xyz_row . kinv_t then . rot
 * (two fp32 3x3 products per pixel, in that order); out (height*width, 2) = (u, v). */
int se3ds_persp_from_equirect_coords(const float* kinv_t, const float* rot, int height, int width,
                                     int eq_h, int eq_w, float* out, void* stream);

/* mask_pano -- utils/pano_utils.py:245-265: rows [mh, H-mh] kept, others := value. */
int se3ds_mask_pano(const void* pano, int dtype, int n, int height, int width, int channels,
                    int masked_height, float value, void* out, void* stream);

/* Stream compaction of valid points -- models/models.py:229-236.  Keeps point j iff
 * any(feats[:, j, :] != void) over batch and channel.  xyz1 (N,4,M), feats (N,M,C) of dtype.
 * Outputs xyz1_out (N,4,M) / feats_out (N,M,C) hold `*count` points per row with the
 * compacted stride `out_stride` (== M); count_out is a DEVICE int64.  workspace:
 * se3ds_compact_workspace_bytes(m). */
size_t se3ds_compact_workspace_bytes(int64_t m);
int se3ds_compact_valid(const float* xyz1, const void* feats, int feat_dtype, int n, int64_t m,
                        int channels, float void_class, float* xyz1_out, void* feats_out,
                        int64_t* count_out, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SE3DS_HIP_H_ */
