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
/* OR-ed into the `feat_dtype` of the splat entry points (se3ds_project_equirect*,
 * se3ds_project_to_feat) together with SE3DS_I32: the caller PROMISES that every feature of a
 * valid point -- one whose channels all differ from `input_void` -- is an integer in [0, 255]
 * (RGB memories: int32 in [-1, 255] with void -1, reference models/models.py:127-134,325-331).
 * The splat then moves 8-byte packed records instead of 20-byte ones.  SE3DS_U8 features need no
 * promise.  A broken promise is detected (se3ds_splat_promise_broken) but the outputs of that
 * call are undefined. */
#define SE3DS_FEAT_BYTE_RANGE 0x100
/* OR-ed into the `feat_dtype` of se3ds_unproject_equirect_into / se3ds_warp_views_to_target: the caller
 * guarantees that row 3 of xyz1 (the homogeneous 1, models/models.py:225-226 adds 0 to it) ALREADY holds
 * 1.0 in the window being written -- a point-cloud memory fills that row once when it is allocated -- and
 * the kernel does not write it again: 40 instead of 44 bytes per pixel (round 6; the splat never reads
 * the row, the memory keeps it for callers of the (N,4,M) view). */
#define SE3DS_XYZ1_ONES_PRESET 0x200

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

/* The same, written straight into a point-cloud MEMORY (the concat of models/models.py:239-245
 * and utils/eval_metric.py:238-239 without the copy): xyz1 is (N,4,m_total) and feats_out
 * (N,m_total,C); this view fills columns / rows [m_offset, m_offset + H*W). */
int se3ds_unproject_equirect_into(const void* feats, int feat_dtype, const float* depth,
                                  const float* sin_el, const float* cos_el, const float* sin_hd,
                                  const float* cos_hd, const float* position, int n, int height,
                                  int width, int channels, float void_class, float depth_scale,
                                  float* xyz1, void* feats_out, int64_t m_total, int64_t m_offset,
                                  void* stream);

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

/* Number of elements of an int32 / uint8 feature array that are neither `void_class` nor an
 * integer in [0, 255]: 0 means SE3DS_FEAT_BYTE_RANGE may be promised for it.  *bad_out is a
 * device uint32 (written, not accumulated). */
int se3ds_feats_byte_range(const void* feats, int feat_dtype, int64_t count, float void_class,
                           uint32_t* bad_out, void* stream);
/* After a packed / sorted splat call (same workspace, same n / m): *broken_out (device uint32) = 1
 * if a valid point of THAT call violated the SE3DS_FEAT_BYTE_RANGE promise, else 0 (the verdict is
 * an epoch the violating workgroups exchange into the workspace: nothing is zeroed per call). */
int se3ds_splat_promise_broken(const void* workspace, int n, int64_t m, uint32_t* broken_out,
                               void* stream);
/* The same verdict, STICKY over calls: header word 3 of a splat workspace (byte 12) is OR-ed with 1
 * by every packed / sorted splat that meets a feature outside [0, 255] under the promise, and is
 * never cleared by the library.  The caller zeroes the first 256 bytes of a workspace once, and
 * reads the flag whenever convenient (e.g. every n-th call, asynchronously): a broken promise is
 * then detected late, never missed.  clear != 0 resets the word after reading it. */
int se3ds_splat_promise_sticky(void* workspace, uint32_t* broken_out, int clear, void* stream);

/* The same over the first `m` points of a preallocated point-cloud MEMORY of `capacity` points
 * per image -- xyz1 (N,4,capacity), feats (N,capacity,C) -- so that a trajectory appends frames in
 * place (se3ds_unproject_equirect_into) and renders without ever concatenating or copying the
 * memory (utils/eval_metric.py:162-165,236-239; trainers/gan_manager.py:476-485,540-541). */
int se3ds_project_equirect_memory(const float* xyz1, const float* offset, const void* feats,
                                  int feat_dtype, int n, int64_t m, int64_t capacity, int channels,
                                  int height, int width, float depth_scale, float input_void,
                                  float output_void, float* depth, float* feat, float* mask,
                                  float mask_void, void* workspace, size_t workspace_bytes,
                                  void* stream);

/* One trajectory step as ONE host call: `views` source panoramas are unprojected into consecutive
 * windows [m_offset + v*H*W, ...) of a point-cloud memory (se3ds_unproject_equirect_into each, with
 * view_position[v] (N,3) added, or view_position == NULL) and ONE target is rendered from the
 * memory's first m_offset + views*H*W points, relative to `target` (N,3)
 * (se3ds_project_equirect_memory) -- utils/eval_metric.py:153-166,233-240 and the RE10K notebook's
 * cell 15 per step, trainers/gan_manager.py:476-485,540-541.  view_feats / view_depth /
 * view_position are HOST arrays of `views` device pointers.  The launches are queued back to
 * back: from Python the four separate calls of a 2-view step cost 154 us of host time for 124 us
 * of kernels.  feat_dtype may carry SE3DS_FEAT_BYTE_RANGE (a promise about every view's features
 * AND what the memory already holds).  views == 0 renders the memory as it is. */
int se3ds_warp_views_to_target(const void* const* view_feats, int feat_dtype,
                               const float* const* view_depth, const float* const* view_position,
                               int views, int n, int height, int width, int channels,
                               float void_class, float depth_scale, const float* sin_el,
                               const float* cos_el, const float* sin_hd, const float* cos_hd,
                               float* mem_xyz1, void* mem_feats, int64_t capacity, int64_t m_offset,
                               const float* target, int out_height, int out_width,
                               float output_void, float* depth, float* feat, float* mask,
                               float mask_void, void* workspace, size_t workspace_bytes,
                               void* stream);

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

/* Debug/parity tap: the DEVICE evaluation of the fast index screen (se3ds_geom_math.h) on
 * camera-relative xyz (3,M): fx, fy as the screen computes them and its verdict per point
 * (>= -1: decided index v*W+u or -1 = not a valid splat; -2: left to the exact chain). */
int se3ds_debug_fast_fxy(const float* xyz, int64_t m, int width, int height, float* fx, float* fy,
                         int32_t* verdict, void* stream);

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

/* Fused perspective paths (notebooks/SE3DS_RE10K_Colab.ipynb cells 15, 17).
 * se3ds_perspective_to_pointcloud = project_perspective_image(image) and (depth)
 * (utils/pano_utils.py:344-417, constant padding `pad_value`), int32(image * 255), and
 * equirectangular_to_pointcloud (:164-242, + optional position (3)) without the two equirect
 * intermediates: image (ih,iw,C<=4) fp32, depth (ih,iw) fp32, rays (3,H*W) = equirectangular
 * pixel rays, w2i (3,3) = get_world_to_image_transform; outputs xyz1 (1,4,H*W), feats (1,H*W,C)
 * int32.  se3ds_perspective_guidance = the three get_perspective_from_equirectangular_image
 * calls (:443-476) of cell 17 on the splat outputs pred_rgb (H,W,3) / pred_depth (H,W) -- RGB,
 * depth, and the mask (depth != 0, != 1, all(rgb != 0)) -- with the glue: rgb / 255 clipped,
 * mask == 1, products with the mask: proj_image (h,w,3), proj_depth (h,w), proj_mask (h,w). */
int se3ds_perspective_to_pointcloud(const float* image, const float* depth, int ih, int iw,
                                    int channels, const float* rays, const float* w2i,
                                    int round_nearest, float pad_value, const float* sin_el,
                                    const float* cos_el, const float* sin_hd, const float* cos_hd,
                                    const float* position, int height, int width, float void_class,
                                    float depth_scale, float* xyz1, int32_t* feats_out, void* stream);
int se3ds_perspective_guidance(const float* pred_rgb, const float* pred_depth, int eq_h, int eq_w,
                               const float* kinv_t, const float* rot, int height, int width,
                               float* proj_image, float* proj_depth, float* proj_mask,
                               void* stream);

/* tf.image.resize with half-pixel centres on an NHWC tensor (F32 / I32 / U8): method 0 = nearest
 * (output of the input dtype), 1 = bilinear (fp32 output, antialias off).  Callers:
 * utils/pano_utils.py:203-208 (equirectangular_to_pointcloud with size_mult != 1) and :299-301
 * (crop_pano(resize_to_original=True): an up-scaling, where TF's antialias=True changes nothing). */
int se3ds_resize(const void* x, int dtype, int n, int h, int w, int c, int oh, int ow, int method,
                 void* y, void* stream);
/* *out = mean(x[0..n)) accumulated in binary64: the padding value of
 * project_perspective_image(pad_mode='mean'), utils/pano_utils.py:403-407. */
int se3ds_mean_f32(const float* x, int64_t n, float* out, void* stream);

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

/* ======================================================================================
 * Convolutions (implicit GEMM on MFMA; NHWC; dtype = SE3DS_F32 or SE3DS_BF16, fp32 accumulate)
 * Geometry arguments always describe the ASSOCIATED FORWARD CONV: input (n,h,w,cin), output
 * (n,ho,wo,cout), kernel kh x kw, stride (1|2), explicit top/left zero padding (the bottom /
 * right padding follows from ho/wo), wrap_w = circular padding along W (PadLayer at
 * inference, models/layers.py:67-72; stride 1 and wo == w only).
 * ====================================================================================== */

/* fp32 master kernel HWIO viewed [K = kh*kw*cin][cout] -> compute-dtype operand copies:
 * wt [cout][K] (forward operand) and, if non-NULL, wn [K][cout] (input-gradient operand). */
int se3ds_weight_prep(const float* w, int64_t k, int cout, int dtype, void* wt, void* wn,
                      void* stream);
/* The same for MANY layers in one launch (bf16 only; every layer needs k % 8 == 0, cout % 4 == 0,
 * 16-byte aligned w / wt, 8-byte aligned wn): `table` is a device int64 [nlayers][6] =
 * {w, k, cout, wt, wn, first 64 x 64 tile}, total_tiles the sum of ceil(k/64) * ceil(cout/64).
 * The trainer refreshes a whole module's operand copies behind its Adam update with it. */
#define SE3DS_WEIGHT_PREP_FIELDS 6
int se3ds_weight_prep_multi(const int64_t* table, int nlayers, int64_t total_tiles, void* stream);

/* y = epilogue(conv(x * in_mask, W)).  Replaces tf.nn.conv2d at models/layers.py:193-198
 * (PartialConv), :334-339 (SpectralConv), Keras Conv2D (image_models.py:513-517,542-543),
 * and is the input-gradient of Conv2DTranspose.  in_mask (n,h,w) fp32 or NULL;
 * in_mask_binary != 0 promises that it only holds 0.0 / 1.0 (the reference's masks are binary),
 * which lets masked pixels be fetched from a zero page by the LDS-DMA kernel.  Epilogue:
 *   t = acc * (*scale)                                  scale: device scalar or NULL
 *   row_a != NULL, bias != NULL: t = ((t - b)*row_a + b) * row_b   (layers.py:199-202)
 *   row_a != NULL, bias == NULL: t = t * row_a                     (layers.py:203-204)
 *   row_a == NULL, bias != NULL: t = t + b
 *   act: 0 none, 1 ReLU, 2 LeakyReLU(act_alpha)
 * row_a / row_b are (n*ho*wo) fp32 vectors (mask_ratio / update_mask). */
int se3ds_conv2d_fwd(const void* x, const void* wt, void* y, int dtype, int n, int h, int w,
                     int cin, int ho, int wo, int cout, int kh, int kw, int stride, int pad_t,
                     int pad_l, int wrap_w, const float* in_mask, int in_mask_binary,
                     const float* scale, const float* bias, const float* row_a,
                     const float* row_b, int act, float act_alpha, void* stream);

/* Keras Conv2DTranspose, 2x2 kernel, stride 2 (layers.py:475-480 residual upsampling;
 * image_models.py:440-441 final_deconv): y (n, 2 hi, 2 wi, cout) from x (n, hi, wi, cin).  `wn` is
 * the compute-dtype copy of the kernel in its own layout (ky, kx, cout, cin) -- se3ds_weight_prep's
 * second output for the layer.  Each output pixel sees exactly one tap, so the layer is two 1x1
 * forward convolutions (one per output row parity) with 2 * cout "channels" (kx, co) whose results
 * are contiguous in y: round 5 replaces the four parity-class passes of se3ds_conv2d_dgrad (K = cin:
 * two K steps per 128 x 128 tile, 256-byte output pieces) by 256-wide tiles writing 512-byte runs.
 * bias (cout floats) or NULL. */
int se3ds_conv_transpose2x2_fwd(const void* x, const void* wn, void* y, int dtype, int n, int hi, int wi,
                                int cin, int cout, const float* bias, void* stream);

/* Forward conv that also emits the batch-norm statistics of its (rounded) output, so that
 * SyncBatchNormalization (models/layers.py BN after conv) needs no separate pass over y:
 * stats[row][2][cout] = per-row-block (sum, sum of squares) over the stored outputs, with
 * rows = se3ds_conv2d_fwd_stats_rows(...) (0: this shape has no fused path, use
 * se3ds_conv2d_fwd + se3ds_norm_stats).  Reduce with se3ds_norm_reduce_rows. */
int64_t se3ds_conv2d_fwd_stats_rows(int dtype, int n, int cin, int ho, int wo, int cout, int kh,
                                    int kw, int stride, int has_in_mask, int in_mask_binary);
int se3ds_conv2d_fwd_stats(const void* x, const void* wt, void* y, int dtype, int n, int h, int w,
                           int cin, int ho, int wo, int cout, int kh, int kw, int stride,
                           int pad_t, int pad_l, int wrap_w, const float* in_mask,
                           int in_mask_binary, const float* scale, const float* bias,
                           const float* row_a, const float* row_b, int act, float act_alpha,
                           float* stats, void* stream);

/* dx = epilogue(conv_transpose(dy, W)): the input-gradient of the conv above AND the forward
 * of Keras Conv2DTranspose (models/layers.py:417-423,475-480; image_models.py:440-441), whose
 * kernel (kh,kw,Cout_T,Cin_T) is the HWIO kernel of the associated forward conv.  wn is the
 * [K][cout] operand copy.  dy_row_scale (n*ho*wo) optionally multiplies dy rows (partial
 * conv: ratio*update_mask).  Epilogue: t = acc*(*scale); row_a (n*h*w): t *= row_a (the
 * partial conv's input mask); bias (cin) added when given (ConvT bias); act as above. */
int se3ds_conv2d_dgrad(const void* dy, const void* wn, void* dx, int dtype, int n, int h, int w,
                       int cin, int ho, int wo, int cout, int kh, int kw, int stride, int pad_t,
                       int pad_l, int wrap_w, const float* dy_row_scale, const float* scale,
                       const float* bias, const float* row_a, int act, float act_alpha,
                       void* stream);
/* The same with dx = result + addend (addend: same layout / dtype as dx, may be dx itself): the
 * gradient of a tensor with two consumers (a ResNet block's input) without an extra pass. */
int se3ds_conv2d_dgrad_acc(const void* dy, const void* wn, void* dx, int dtype, int n, int h, int w,
                       int cin, int ho, int wo, int cout, int kh, int kw, int stride, int pad_t,
                       int pad_l, int wrap_w, const float* dy_row_scale, const float* scale,
                       const float* bias, const float* row_a, int act, float act_alpha,
                       const void* addend, void* stream);

/* Data gradient with FUSED batch-norm backward statistics (round 3).  dx (+ addend when given)
 * is the gradient of y = act(norm(bn_x) [+ res]), the output of a SyncBatchNormalization
 * (tf.keras.layers.experimental.SyncBatchNormalization behind models/layers.py:241-251,
 * 424-444); besides dx the epilogue emits stats[rows][2][cin] = per 64-pixel tile
 * (sum dz, sum dz * xhat) with dz = dx_stored * act'(y) (bn_mask: one "y > 0" bit per element,
 * or NULL without activation; bn_act 0 / 1 relu / 2 leaky relu with slope bn_alpha) and
 * xhat = (bn_x - bn_mean) * bn_rstd -- the sums se3ds_norm_bwd_stats takes from a second pass over
 * dx and bn_x.  se3ds_norm_reduce_rows_dst reduces the rows (and writes the beta / gamma
 * gradients).  _rows returns 0 when this shape cannot (strided, fp32, thin / ragged channels,
 * a dy row scale): call se3ds_conv2d_dgrad[_acc] and se3ds_norm_bwd_stats then. */
int64_t se3ds_conv2d_dgrad_bnstats_rows(int dtype, int n, int h, int w, int cin, int cout, int kh,
                                        int kw, int stride, int has_row_scale);
int se3ds_conv2d_dgrad_bnstats(const void* dy, const void* wn, void* dx, int dtype, int n, int h,
                               int w, int cin, int ho, int wo, int cout, int kh, int kw, int stride,
                               int pad_t, int pad_l, int wrap_w, const float* scale,
                               const float* row_a, const void* addend, const void* bn_x,
                               const uint8_t* bn_mask, const float* bn_mean, const float* bn_rstd,
                               int bn_act, float bn_alpha, float* stats, void* stream);

/* dW[kh,kw,cin,cout] (fp32) (+)= (*out_scale) * sum_pixels (x*in_mask)^T (dy*row_scale).
 * The reduction over n*ho*wo is split across workgroups and reduced deterministically. */
size_t se3ds_conv2d_wgrad_workspace_bytes(int n, int ho, int wo, int cin, int cout, int kh,
                                          int kw);
int se3ds_conv2d_wgrad(const void* x, const void* dy, float* dw, int dtype, int n, int h, int w,
                       int cin, int ho, int wo, int cout, int kh, int kw, int stride, int pad_t,
                       int pad_l, int wrap_w, const float* in_mask, int in_mask_binary,
                       const float* row_scale, const float* out_scale, int accumulate,
                       void* workspace, size_t workspace_bytes, void* stream);
/* The same without its final split reduction (round 4): the partial slabs stay in `workspace`
 * (which the caller therefore keeps per layer until the reduction has run) and reduce_row (HOST
 * memory, 5 x int64) receives [slabs, splits, n / 4, dw, 1]; rows of many layers go to ONE
 * se3ds_wgrad_reduce_multi launch (device table of such rows with the 5th field replaced by the
 * row's first workgroup: running sum of ceil(n4 / se3ds_wgrad_reduce_tile())).  When the 16-byte
 * reduction does not apply (ragged sizes), the reduction is launched as usual and reduce_row[4]
 * stays 0.  Bit-identical to se3ds_conv2d_wgrad.  Same reference semantics: the kernel gradient of
 * tf.nn.conv2d (models/layers.py:153,193,334). */
int se3ds_conv2d_wgrad_partial(const void* x, const void* dy, float* dw, int dtype, int n, int h,
                               int w, int cin, int ho, int wo, int cout, int kh, int kw, int stride,
                               int pad_t, int pad_l, int wrap_w, const float* in_mask,
                               int in_mask_binary, const float* row_scale, void* workspace,
                               size_t workspace_bytes, int64_t* reduce_row, void* stream);
int se3ds_wgrad_reduce_multi(const int64_t* table, int rows, int64_t workgroups, void* stream);
int se3ds_wgrad_reduce_tile(void);
/* n / d as the convolution kernels compute it for tile row -> (image, row, column): the host-made
 * multiplier and shifts, evaluated on the host (test hook: no device work).  d >= 1. */
uint32_t se3ds_fastdiv_host(uint32_t n, uint32_t d);

/* Weight gradient of a THIN-Cout (cout <= 16), stride-1, same-size conv (the generator's
 * 128->3 / 128->1 output convs, image_models.py:93-104) computed with the operand roles
 * swapped: sum_l dy[l - tap, co] * x[l, ci] is the weight gradient of a conv whose INPUT is dy and
 * whose output gradient is x, with the taps mirrored.  The thin operand then packs into one
 * linear-K tile and x is streamed once instead of kh*kw times. */
size_t se3ds_conv2d_wgrad_swapped_workspace_bytes(int n, int h, int w, int cin, int cout, int k);
int se3ds_conv2d_wgrad_swapped(const void* x, const void* dy, float* dw, int dtype, int n, int h,
                               int w, int cin, int cout, int k, int pad, int accumulate,
                               void* workspace, size_t workspace_bytes, void* stream);

/* Partial-conv mask statistics -- models/layers.py:153-163: cnt = window sum of the mask
 * (n,h,w); ratio = kh*kw/(cnt+1e-6)*clip(cnt,0,1); um = clip(cnt,0,1); optionally
 * ru = ratio*um and bu = (1-ratio)*um (backward helpers).  All (n,ho,wo) fp32. */
int se3ds_mask_window(const float* mask, int n, int h, int w, int ho, int wo, int kh, int kw,
                      int stride, int pad_t, int pad_l, int wrap_w, float* ratio, float* um,
                      float* ru, float* bu, void* stream);

/* se3ds_norm_reduce_rows + se3ds_norm_finalize in one launch for the single-replica batch norm
 * whose column statistics came out of the producing convolution's epilogue (rows <= 2048 partial
 * rows of [2][c]); bit-identical to the pair.  Replaces the statistics half of
 * tf.keras.layers.experimental.SyncBatchNormalization (image_models.py:170-176). */
int se3ds_norm_reduce_rows_finalize(const float* partial, int64_t rows, int c, float count,
                                    const float* gamma, const float* beta, float eps, float momentum,
                                    float* moving_mean, float* moving_var, float* scale, float* shift,
                                    float* mean, float* rstd, void* stream);

/* One pass over dy [r][c] (bf16, c % 8 == 0) for the backward of a partial convolution with bias:
 * scaled_out = dy * out_row_scale[row]  (what se3ds_row_scale would write: the renormalised output
 * gradient the LDS-DMA data / weight gradient kernels take) and colsum_out[c] = sum_rows dy *
 * sum_row_scale[row] (the bias gradient, models/layers.py:199-203).  sums: [1][2][c] scratch like
 * se3ds_norm_stats.  SE3DS_E_UNSUPPORTED for other layouts (callers use the two separate calls). */
int se3ds_colsum_row_scale(const void* x, int dtype, int64_t r, int c, const float* sum_row_scale,
                           const float* out_row_scale, void* scaled_out, float* sums,
                           float* colsum_out, void* workspace, size_t workspace_bytes,
                           void* stream);

/* ======================================================================================
 * Normalisation: tensors viewed as [g][r][c]; g = 1 for SyncBatchNormalization (279 sites in
 * the generator, e.g. models/layers.py:235-251), g = batch for tfa InstanceNormalization
 * (image_models.py:534).  sums are [g][2][c] fp32.
 * ====================================================================================== */
size_t se3ds_norm_workspace_bytes(int g, int c);
/* sums[g][0][c] = sum_r x*row_scale, sums[g][1][c] = sum_r (x*row_scale)^2 (row_scale may be NULL);
 * also the column-sum primitive behind bias gradients: colsum_out (c floats, may be NULL)
 * additionally receives sums[0][0][:] (written straight into a gradient arena). */
int se3ds_norm_stats(const void* x, int dtype, int g, int64_t r, int c, const float* row_scale,
                     float* sums, float* colsum_out, void* workspace, size_t workspace_bytes,
                     void* stream);
/* sums[2][c] = sum over rows of partial[rows][2][c] (deterministic; workspace of
 * se3ds_norm_workspace_bytes(ceil(rows/512), c) bytes for rows > 2048). */
int se3ds_norm_reduce_rows(const float* partial, int64_t rows, int c, float* sums, void* workspace,
                           size_t workspace_bytes, void* stream);
/* The same; dst0 / dst1 (c floats each, may be NULL) additionally receive sums[0][:] / sums[1][:]
 * (the beta / gamma gradients when the rows are fused batch-norm backward statistics). */
int se3ds_norm_reduce_rows_dst(const float* partial, int64_t rows, int c, float* sums, float* dst0,
                               float* dst1, void* workspace, size_t workspace_bytes, void* stream);
/* mean = S0/count, var = S1/count - mean^2 (biased); scale = gamma*rsqrt(var+eps),
 * shift = beta - mean*scale; moving stats (c) updated in place unless NULL;
 * use_moving != 0: inference (statistics from moving_mean / moving_var). */
int se3ds_norm_finalize(const float* sums, float count, int g, int c, const float* gamma,
                        const float* beta, float eps, float momentum, float* moving_mean,
                        float* moving_var, int use_moving, float* scale, float* shift, float* mean,
                        float* rstd, void* stream);
/* y = act(x*scale + shift [+ res]) [+ post]  (scale/shift are [g][c]).
 * act_mask (bf16, c % 8 == 0; may be NULL): g*r*c/8 bytes, bit e of byte i = (y[8*i + e] > 0).
 * The backward entry points below take the same buffer and then do not read y (16x less
 * traffic for the activation derivative); with act_mask == NULL they read y. */
int se3ds_norm_apply(const void* x, int dtype, int g, int64_t r, int c, const float* scale,
                     const float* shift, const void* res, const void* post, int act, float alpha,
                     void* y, void* act_mask, void* stream);
/* sums[g][0][c] = sum dpre, sums[g][1][c] = sum dpre*xhat with dpre = dy*act'(y).  For g == 1
 * dbeta_out / dgamma_out (c floats, may be NULL) receive the two rows directly. */
int se3ds_norm_bwd_stats(const void* dy, const void* y, const void* x, int dtype, int g, int64_t r,
                         int c, const float* mean, const float* rstd, int act, float alpha,
                         float* sums, float* dbeta_out, float* dgamma_out, const void* act_mask,
                         void* workspace, size_t workspace_bytes, void* stream);
/* dx = gamma*rstd*(dpre - S0/count - xhat*S1/count); dres = dpre when non-NULL.
 * in_act != 0: x is itself the output of activation in_act (alpha in_alpha) whose producer
 * skips its own derivative pass: dx is additionally multiplied by act'(x). */
int se3ds_norm_bwd_apply(const void* dy, const void* y, const void* x, int dtype, int g, int64_t r,
                         int c, const float* mean, const float* rstd, const float* gamma,
                         const float* sums, float count, int act, float alpha, void* dx,
                         void* dres, const void* act_mask, int in_act, float in_alpha,
                         void* stream);
/* se3ds_norm_bwd_apply for a SyncBatchNormalization (one group, bf16, c % 8 == 0) whose input x
 * is the output of a PartialConv with a bias (models/layers.py:100-209) and this norm its only
 * consumer: dx is stored pre-scaled by out_row[row] (mask_ratio * update_mask, the multiplier of
 * layers.py:199-204 the conv's weight / data gradients need) and colsum_dst[c] = sum over rows of
 * the rounded dx * sum_row[row] (update_mask: the bias gradient) -- what se3ds_colsum_row_scale
 * takes from a pass of its own over dx.  workspace: se3ds_norm_workspace_bytes(3, c).
 * SE3DS_E_UNSUPPORTED where the fast kernel does not apply. */
int se3ds_norm_bwd_apply_rows(const void* dy, const void* x, int dtype, int64_t r, int c,
                              const float* mean, const float* rstd, const float* gamma,
                              const float* sums, float count, int act, float alpha, void* dx,
                              void* dres, const void* act_mask, const float* sum_row,
                              const float* out_row, float* colsum_dst, void* workspace,
                              size_t workspace_bytes, void* stream);
/* Batch-norm backward (one group, bf16) in TWO launches instead of three (round 5): statistics
 * partials + apply in the channel-group layout (64 channels = one 128-byte line per row and
 * workgroup); every apply workgroup folds the <= 64 partial rows of its own channels in a prologue,
 * the stand-alone column reduction between se3ds_norm_bwd_stats and se3ds_norm_bwd_apply is gone
 * (layers.py:235-251: the backward of every BatchNormalization of the generator).  dbeta_out /
 * dgamma_out (c floats, or NULL) receive sum dz / sum dz * xhat, sums_out ([2][c] or NULL) both.
 * With sum_row / out_row (the rows variant, se3ds_norm_bwd_apply_rows): dx is stored pre-scaled and
 * colpart[se3ds_norm_bwd_cg_col_rows(r, c)][c] receives the bias-gradient partials as compact slabs
 * -- reduce them with se3ds_wgrad_reduce_multi (row: colpart, rows, c / 4, destination).
 * workspace: se3ds_norm_bwd_cg_workspace_bytes(c).  SE3DS_E_UNSUPPORTED where
 * se3ds_norm_bwd_cg_supported() is 0 (c % 64, c < 512, other dtypes; SE3DS_NORM_CG=0). */
int se3ds_norm_bwd_cg_supported(int dtype, int64_t r, int c, int act, int has_mask, int in_act);
size_t se3ds_norm_bwd_cg_workspace_bytes(int c);
int se3ds_norm_bwd_cg_col_rows(int64_t r, int c);
int se3ds_norm_bwd_cg(const void* dy, const void* x, int dtype, int64_t r, int c, const float* mean,
                      const float* rstd, const float* gamma, float count, int act, float alpha,
                      void* dx, void* dres, const void* act_mask, int in_act, float in_alpha,
                      float* dbeta_out, float* dgamma_out, float* sums_out, const float* sum_row,
                      const float* out_row, float* colpart, void* workspace, size_t workspace_bytes,
                      void* stream);
/* inference-mode backward: dx = dpre*scale; dres = dpre. */
int se3ds_affine_bwd(const void* dy, const void* y, int dtype, int g, int64_t r, int c,
                     const float* scale, int act, float alpha, void* dx, void* dres,
                     const void* act_mask, void* stream);

/* ======================================================================================
 * Pointwise / pooling
 * ====================================================================================== */
int se3ds_act_bwd(const void* dy, const void* y, int dtype, int64_t n, int act, float alpha,
                  void* dx, void* stream);
int se3ds_add(const void* a, const void* b, int dtype, int64_t n, void* out, void* stream);
/* out[r,:] = x[r,:] * scale[r] (partial-conv backward: dy * ratio * update_mask). */
int se3ds_row_scale(const void* x, int dtype, int64_t rows, int c, const float* scale, void* out,
                    void* stream);
/* Keras MaxPool2D(2, padding='SAME') -- image_models.py:267,289 */
int se3ds_maxpool2x2_fwd(const void* x, int dtype, int n, int h, int w, int c, void* y,
                         void* stream);
int se3ds_maxpool2x2_bwd(const void* dy, const void* x, const void* y, int dtype, int n, int h,
                         int w, int c, void* dx, void* stream);
/* tf.nn.avg_pool(ksize=3, strides=2, 'SAME') -- image_models.py:617 */
int se3ds_avgpool3s2_fwd(const void* x, int dtype, int n, int h, int w, int c, void* y,
                         void* stream);
int se3ds_avgpool3s2_bwd(const void* dy, int dtype, int n, int h, int w, int c, void* dx,
                         void* stream);
/* Keras UpSampling2D() nearest x2 -- image_models.py:371,378 */
int se3ds_upsample2x_fwd(const void* x, int dtype, int n, int h, int w, int c, void* y,
                         void* stream);
int se3ds_upsample2x_bwd(const void* dy, int dtype, int n, int h, int w, int c, void* dx,
                         void* stream);
/* dst[r, dst_c0 + i] = convert(src[r, src_c0 + i]), i < ncopy: tf.concat / tf.split along
 * channels with dtype conversion (image_models.py:157-162, se3ds_trainer.py:181-192). */
int se3ds_copy_channels(const void* src, int src_dtype, int src_c, int src_c0, void* dst,
                        int dst_dtype, int dst_c, int dst_c0, int ncopy, int64_t rows,
                        void* stream);
int se3ds_fill(void* p, int dtype, int64_t n, float value, void* stream);
/* SE3DSModel quantisation glue -- models/models.py:198 (int / 255 -> fp32), :289-291
 * (proj_rgb / 255 clipped to [0,1]), :325-331 (int32(g * 255) clipped to [-1,255];
 * int32(clip(g,0,1) * 255)), :353,:359 (casts to uint8): out = Q(clamp?(in) * mul / div) with one
 * rounding per op; float outputs are clamped to [lo,hi] as tf.clip_by_value does (NaN
 * propagates), integer outputs truncate toward zero (tf.cast) and are then clamped to [lo,hi];
 * lo > hi selects the pure cast (no clamp; int32 -> uint8 wraps modulo 256 like tf.cast).
 * dtypes: SE3DS_F32 / SE3DS_I32 / SE3DS_U8. */
int se3ds_quantize(const void* in, int in_dtype, int64_t n, int pre_clamp, float pre_lo,
                   float pre_hi, float mul, float div, float lo, float hi, void* out, int out_dtype,
                   void* stream);
/* PadLayer as a standalone op -- models/layers.py:22-97: pad H and W by `pad`; mode 0
 * CONSTANT(value) / 1 REFLECT / 2 SYMMETRIC; wrap_w != 0: W is padded circularly. */
int se3ds_pad2d(const void* x, int dtype, int n, int h, int w, int c, int pad, int mode, int wrap_w,
                float value, void* y, void* stream);
/* heads -- image_models.py:187-190. kind 0: (tanh(x)+1)/2, kind 1: clip(x,0,1); y is fp32. */
int se3ds_head_fwd(const void* x, int dtype, int64_t n, int kind, float* y, void* stream);
int se3ds_head_bwd(const float* dy, const float* y, const void* x, int dtype, int64_t n, int kind,
                   void* dx, void* stream);

/* ======================================================================================
 * Losses -- trainers/se3ds_trainer.py:39-71,148-234 (fp32)
 * ====================================================================================== */
/* out[n] = per-sample sum; mode 0: sum(a); 1: sum(|a-b|*m[p]); 2: count(0<a<1);
 * 3: sum(a*(1-b)); 4: sum((a-b)^2 * 1[0<b<1]) (depth RMSE numerator, utils/eval_metric.py:225-231).
 * a,b: (n,p,c); m: (n,p). */
int se3ds_sample_sum(const float* a, const float* b, const float* m, int n, int64_t p, int c,
                     int mode, float* out, float* workspace /* n*256 floats */, void* stream);
/* grad = coef[n]*sign(a-b)*w; mode 0: w = 1[0<b<1] (depth L1); mode 1: w = m*(1-m2) (wc). */
int se3ds_l1_grad(const float* a, const float* b, const float* m, const float* m2,
                  const float* coef, int n, int64_t p, int c, int mode, float* grad, void* stream);
/* out[i] = scale / max(sums[i], 1): per-sample loss normalisers (se3ds_trainer.py:151-152). */
int se3ds_recip_clamp(const float* sums, int n, float scale, float* out, void* stream);
/* logits (2*half) [fake | real]: sums[0] = sum(-fake), sums[1] = sum(relu(1-real)+relu(1+fake));
 * dlog_d = cd * d(sums[1])/dlogits, dlog_g = cg * d(sums[0])/dlogits (either may be NULL). */
int se3ds_hinge(const void* logits, int dtype, int64_t half, float cd, float cg, float* sums,
                void* dlog_d, void* dlog_g, void* stream);

/* ======================================================================================
 * Optimiser step over flat fp32 arenas -- se3ds_trainer.py:27-32,238-257;
 * gan_manager.py:175-183; utils/ema.py:54-64
 * ====================================================================================== */
/* chunks: int64 triples (tensor id, start, length<=65536) covering the arena;
 * tensor_chunk_start: (ntensors+1) prefix into chunks.  sqnorm[t] = sum g^2 of tensor t. */
int se3ds_multi_sqnorm(const float* grads, const int64_t* chunks, int64_t nchunks,
                       const int64_t* tensor_chunk_start, int ntensors, float* partial,
                       float* sqnorm, void* stream);
/* tf.clip_by_norm per tensor, in place: g = (g*clip)/max(||g||, clip); mean_norm_out (device
 * scalar, may be NULL) = mean over tensors of the clipped norms (NaN -> 0). */
int se3ds_multi_clip_by_norm(float* grads, const int64_t* chunks, int64_t nchunks,
                             const float* sqnorm, int ntensors, float clip_norm,
                             float* mean_norm_out, void* stream);
/* out[0] = mean over tensors of min-clipped norms (the grad_norm metrics, se3ds_trainer.py:239). */
int se3ds_mean_clipped_norm(const float* sqnorm, int ntensors, float clip_norm, float* out,
                            void* stream);
/* Keras Adam (ResourceApplyAdam), `step` = iteration count t >= 1. */
int se3ds_multi_adam_keras(float* params, const float* grads, float* m, float* v, int64_t n,
                           float lr, float beta1, float beta2, float eps, int64_t step,
                           void* stream);
/* The same, and the EMA copy of the parameters advanced in the same pass (ema may be NULL):
 * ema -= (ema - params_new) * one_minus_decay  (utils/ema.py:54-88 applied to the trainable
 * variables right where they are produced). */
int se3ds_multi_adam_keras_ema(float* params, const float* grads, float* m, float* v, int64_t n,
                               float lr, float beta1, float beta2, float eps, int64_t step,
                               float* ema, float one_minus_decay, void* stream);
/* One replica: se3ds_multi_clip_by_norm_sn + se3ds_multi_adam_keras_ema in ONE pass over the chunk
 * rows [tensor id, first element, length] -- the clipped (and, for spectral kernels, fixed-up)
 * gradient is formed in registers and consumed by the Adam (+ EMA) update; the gradient arena is
 * read once and NOT rewritten (trainers/se3ds_trainer.py:27-32,238-257 with a single replica:
 * clip_by_norm feeds apply_gradients directly).  Bit-identical to the two separate passes.
 * sqnorm from se3ds_multi_sqnorm_sn; chunk rows carry absolute tensor ids and arena offsets; all
 * arenas 16-byte aligned (SE3DS_E_UNSUPPORTED otherwise).  ema may be NULL. */
int se3ds_multi_clip_adam_keras_ema(float* params, const float* grads, float* m, float* v,
                                    const int64_t* chunks, int64_t nchunks, const float* sqnorm,
                                    float clip_norm, const int64_t* tensor_sn, float lr, float beta1,
                                    float beta2, float eps, int64_t step, float* ema,
                                    float one_minus_decay, void* stream);
/* ema -= (ema - vars) * one_minus_decay */
int se3ds_multi_ema(float* ema, const float* vars, int64_t n, float one_minus_decay, void* stream);

/* Spectral normalisation, batched over layers -- models/layers.py:312-331.  `table` is a
 * device array of se3ds_spectral_table_fields() int64 per layer:
 *   W, u, v (K), uhat (cout), sig (2: sigma, 1/(sigma+1e-10)), part (part_rows*cout),
 *   vpart (vpart_len), K, cout, grad
 * power_iter: v_hat, u_hat, sigma from (W,u); u := u_hat when training != 0.
 * bwd_fixup: grad := grad/(sigma+eps) - <grad,W>/(sigma+eps)^2 * v_hat^T u_hat. */
int se3ds_spectral_table_fields(void);
int se3ds_spectral_part_rows(void);
int se3ds_spectral_vpart_len(void);
int se3ds_spectral_power_iter(const int64_t* table, int nlayers, int training, void* stream);
int se3ds_spectral_bwd_fixup(const int64_t* table, int nlayers, void* stream);
/* The fix-up folded into the clip pass (one read-modify-write of the gradient arena instead of
 * two, no separate norm pass over the spectral kernels):
 *   se3ds_spectral_bwd_dots       pass 1 only: <grad,W>, <grad,grad>, <grad, v_hat^T u_hat> into vpart
 *   se3ds_multi_sqnorm_sn         as se3ds_multi_sqnorm; tensors t with tensor_sn[t] != 0 (address of
 *                                 their table row) get |fixed grad|^2 in closed form from vpart
 *   se3ds_multi_clip_by_norm_sn   as se3ds_multi_clip_by_norm; those tensors are fixed up and
 *                                 clipped in the same pass
 * tensor_sn: int64[number of tensors], absolute tensor ids as in the chunk rows; tensor_base: id of
 * the first tensor of tensor_chunk_start / sqnorm (per-segment calls).  Reference semantics:
 * models/layers.py:299-347 (gradient through sigma), trainers/se3ds_trainer.py:27-32 (clip). */
int se3ds_spectral_bwd_dots(const int64_t* table, int nlayers, void* stream);
int se3ds_multi_sqnorm_sn(const float* grads, const int64_t* chunks, int64_t nchunks,
                          const int64_t* tensor_chunk_start, int ntensors, float* partial,
                          float* sqnorm, const int64_t* tensor_sn, int tensor_base, void* stream);
int se3ds_multi_clip_by_norm_sn(float* grads, const int64_t* chunks, int64_t nchunks,
                                const float* sqnorm, int ntensors, float clip_norm,
                                float* mean_norm_out, const int64_t* tensor_sn, void* stream);

/* ======================================================================================
 * Input pipeline, device-side half
 * ====================================================================================== */

/* datasets/indoor_datasets.py:263-375 (_transform_fn) + :553-597 (_train_batch_transform_fn) on
 * decoded frames resident in HBM: convert_image_dtype (:185-228), band masking of proj_mask
 * (:281-304), bilinear (image) / nearest (everything else) resize to (rh, rw) (:312-314), roll by
 * `roll` and optional flip along W (:34-61), crop at (crop_y, crop_x) to (h, w) (:326-330),
 * proj_image *= proj_mask and proj_depth *= proj_mask (:577-585).  Raw frames are (N,H0,W0[,3]);
 * iparams [N][8] = rh, rw, roll, flip, crop_y, crop_x, hmode (0 none / 1 start<x<end / 2 x>start
 * or x<end), vmode (0 / 1); fparams [N][4] = hstart, hend, vstart, vend (raw-grid pixels).
 * Outputs are the step's batch dict entries, fp32 (N,h,w,C) and int32 segmentation. */
int se3ds_input_transform(const uint8_t* image, const uint8_t* proj_image, const uint16_t* depth,
                          const uint16_t* proj_depth, const uint8_t* proj_mask,
                          const uint8_t* blurred_mask, const uint8_t* segmentation,
                          const int32_t* iparams, const float* fparams, int n, int h0, int w0, int h,
                          int w, float* o_image, float* o_proj_image, float* o_proj_mask,
                          float* o_proj_depth, float* o_depth, float* o_blurred, int32_t* o_seg,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SE3DS_HIP_H_ */
