"""Utils for panorama processing -- MI355X implementation of the reference's
utils/pano_utils.py: same function names, argument meaning and error behaviour, torch CUDA
tensors instead of tf.Tensor.  Arithmetic runs in libse3ds_hip.so (se3ds_amd/csrc/geom.hip);
only O(H+W) tables and 3x3 matrices are prepared on the host."""
import math
from typing import Optional, Tuple

import numpy as np
import torch

from se3ds_amd import _lib
from se3ds_amd.utils import _host_tables
from se3ds_amd.utils import point_cloud_utils

F32 = np.float32


def get_world_to_image_transform(image_shape, fov, camera_intrinsics=None, rotations=None,
                                 rotation_matrix=None) -> torch.Tensor:
  """3x3 world->image transform (reference :26-89).  Host-side (nine numbers), fp32 ops."""
  def _np(x):
    return None if x is None else np.asarray(
        x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x, F32)
  fov, camera_intrinsics, rotations, rotation_matrix = (
      _np(fov), _np(camera_intrinsics), _np(rotations), _np(rotation_matrix))
  tan32 = lambda a: F32(np.tan(np.float64(a)))
  if camera_intrinsics is None:
    height, width = F32(image_shape[0]), F32(image_shape[1])
    fov_y, fov_x = fov[0], fov[1]
    fx = F32(F32(0.5) * (width - F32(1.0))) / tan32(fov_x / F32(2))
    fy = F32(F32(0.5) * (height - F32(1.0))) / tan32(fov_y / F32(2))
    camera_intrinsics = np.array([[fx, 0, F32(0.5) * (width - F32(1))],
                                  [0, fy, F32(0.5) * (height - F32(1))], [0., 0, 1]], F32)
  if rotations is not None:
    rp, rh = rotations[0], rotations[1]
    s, c = _host_tables.sin32, _host_tables.cos32
    pitch_rotation = np.array([[1., 0, 0], [0, c(-rp), -s(-rp)], [0, s(-rp), c(-rp)]], F32)
    heading_rotation = np.array([[c(-rh), 0, s(-rh)], [0., 1, 0], [-s(-rh), 0, c(-rh)]], F32)
    extrinsics = np.matmul(pitch_rotation, heading_rotation).astype(F32)
  elif rotation_matrix is not None:
    extrinsics = rotation_matrix
  else:
    extrinsics = np.eye(3, dtype=F32)
  return torch.from_numpy(np.matmul(camera_intrinsics, extrinsics).astype(F32))


def equirectangular_pixel_rays(output_height, device=None) -> torch.Tensor:
  """Unit-ball xyz per equirect pixel, x-right y-down z-forward (reference :92-114).
  Returns (3, H*W) fp32."""
  rays = _host_tables.pixel_rays_np(int(output_height))
  t = torch.from_numpy(rays)
  return t if device is None else t.to(device)


def project_feats_to_equirectangular(feats: torch.Tensor, xyz1: torch.Tensor, height: int,
                                     width: int, void_class: float, depth_scale: float,
                                     offset: Optional[torch.Tensor] = None,
                                     with_mask: bool = False, mask_void: float = -1):
  """Project point-cloud feats into an equirect image (reference :117-161).

  feats (N,M) or (N,M,C); xyz1 (N,4,M).  Returns depth (N,H,W) in [0,1] and feats
  (N,H,W[,C]) fp32.  Extensions (fused, optional): `offset` (N,3) is subtracted from the
  coordinates first (callers' `memory - position`, models.py:273-275); `with_mask` also
  returns models.py:282-287's proj_mask (N,H,W)."""
  return point_cloud_utils._splat('equirect', xyz1, offset, feats, height, width, depth_scale,
                                  void_class, 0, with_mask=with_mask, mask_void=mask_void)


def equirectangular_to_pointcloud(feats: torch.Tensor, depth: torch.Tensor, void_class: float,
                                  depth_scale: float, size_mult: float = 1.0,
                                  interpolation_method: str = 'nearest',
                                  position: Optional[torch.Tensor] = None,
                                  out: Optional[Tuple[torch.Tensor, torch.Tensor, int]] = None,
                                  ones_preset: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
  """Equirect image + depth -> point cloud (reference :164-242).

  feats (N,H,W) or (N,H,W,C) [uint8 / int32 / float32], depth (N,H,W) in [0,1].  Returns xyz1
  (N,4,H*W) and filtered feats (N,H*W[,C]); invalid-depth pixels get xyz (0,0,0,1) and
  void_class.  `position` (N,3), optional, is added to xyz (models.py:225-226).  `out` =
  (memory_xyz1 (N,4,M), memory_feats (N,M,C), offset): write the points straight into columns
  [offset, offset + H*W) of a preallocated memory instead of new tensors (the concat of
  models.py:239-245 without the copy); returns views of the written windows."""
  if feats.dim() != 3 and feats.dim() != 4:
    raise ValueError('feats should have shape (N, H, W) or (N, H, W, C),'
                     f' got {tuple(feats.shape)} instead.')
  if void_class < 0.0 and feats.dtype in (torch.uint8,):
    raise ValueError('feats datatype must be signed if the void class is negative')
  _lib.require_cuda(feats, depth, position)
  is_scalar = feats.dim() == 3
  if is_scalar:
    feats = feats[..., None]
  n, h, w, c = feats.shape
  assert w == 2 * h, 'Expected equirectangular input images'
  if feats.dtype not in (torch.float32, torch.int32, torch.uint8):
    raise ValueError(f'unsupported feats dtype {feats.dtype}')
  if size_mult != 1.0:
    # reference :203-208: depth is resized 'nearest', feats with `interpolation_method`
    sh, sw = int(h * size_mult), int(w * size_mult)
    depth = resize(depth.to(torch.float32)[..., None], sh, sw, 'nearest')[..., 0]
    feats = resize(feats, sh, sw, interpolation_method)
    h, w = sh, sw
  elif interpolation_method != 'nearest' and feats.dtype != torch.float32:
    feats = feats.to(torch.float32)  # bilinear tf.image.resize returns fp32 (identity at 1.0)
  feats = feats.contiguous()
  depth = depth.to(torch.float32).contiguous()
  dev = depth.device
  tab = _host_tables.equirect_tables(h, w, dev)
  base = tab.data_ptr()
  if position is not None:
    position = position.to(torch.float32).contiguous()
  if out is not None:
    mem_x, mem_f, off = out
    m_total = mem_x.shape[2]
    if (mem_x.dtype != torch.float32 or not mem_x.is_contiguous() or not mem_f.is_contiguous() or
        mem_f.dtype != feats.dtype or tuple(mem_x.shape[:2]) != (n, 4) or
        tuple(mem_f.shape) != (n, m_total, c) or off < 0 or off + h * w > m_total):
      raise ValueError('out = (xyz1 (N,4,M) fp32, feats (N,M,C) of the input dtype, offset)')
    xyz1, res = mem_x, mem_f
  else:
    m_total, off = h * w, 0
    xyz1 = torch.empty((n, 4, h * w), dtype=torch.float32, device=dev)
    res = torch.empty((n, h * w, c), dtype=feats.dtype, device=dev)
  # ones_preset (with `out`): row 3 of the window already holds 1.0 -- PointCloudMemory fills it when it
  # allocates -- and the kernel skips those 4 of its 44 bytes per pixel (SE3DS_XYZ1_ONES_PRESET)
  flag = point_cloud_utils.XYZ1_ONES_PRESET if (ones_preset and out is not None) else 0
  rc = _lib.lib().se3ds_unproject_equirect_into(
      _lib.ptr(feats), _lib.dtype_code(feats) | flag, _lib.ptr(depth), base, base + 4 * h,
      base + 8 * h, base + 8 * h + 4 * w, _lib.ptr(position), n, h, w, c, float(void_class),
      float(depth_scale), _lib.ptr(xyz1), _lib.ptr(res), m_total, off, _lib.stream())
  _lib.check(rc, 'se3ds_unproject_equirect')
  if res.dtype == torch.int32:
    # the written features are the source's or void_class: the 8-byte splat's byte-range promise
    # carries over from the source (checked once per source tensor) to the written tensor; a
    # memory filled window by window keeps it only while every window had it
    ok = point_cloud_utils.byte_range(feats, void_class)
    if out is not None:
      prev = getattr(res, '_se3ds_byte_range', None)
      same = (prev is not None and prev[1:4] == (res.data_ptr(), res._version, tuple(res.shape)) and
              prev[4] == float(void_class))
      ok = ok and (prev[0] if same else off == 0)
    point_cloud_utils.set_byte_range(res, ok, void_class)
    if out is None:
      point_cloud_utils.propagate_int_range(res, feats, extra=(void_class,))
  if out is not None:
    xyz1, res = xyz1[:, :, off:off + h * w], res[:, off:off + h * w]
  if is_scalar:
    res = res[..., 0]
  return xyz1, res


def resize(images: torch.Tensor, height: int, width: int, method: str = 'bilinear') -> torch.Tensor:
  """tf.image.resize(images, (height, width), method) on (N,H,W,C), half-pixel centres:
  'nearest' keeps the dtype, 'bilinear' returns fp32 (as TF does).  uint8 / int32 / float32."""
  _lib.require_cuda(images)
  if method not in ('nearest', 'bilinear'):
    raise NotImplementedError(f'resize method {method!r} (nearest / bilinear only)')
  if images.dtype not in (torch.float32, torch.int32, torch.uint8):
    raise ValueError(f'unsupported dtype {images.dtype}')
  images = images.contiguous()
  n, h, w, c = images.shape
  bil = method == 'bilinear'
  if (h, w) == (height, width):
    return images.to(torch.float32) if bil else images
  out = torch.empty((n, height, width, c), dtype=torch.float32 if bil else images.dtype,
                    device=images.device)
  rc = _lib.lib().se3ds_resize(_lib.ptr(images), _lib.dtype_code(images), n, h, w, c, height, width,
                               1 if bil else 0, _lib.ptr(out), _lib.stream())
  _lib.check(rc, 'se3ds_resize')
  return out


def mask_pano(pano: torch.Tensor, proportion: float = 0.125, masked_region_value=0):
  """Masks the top and bottom `proportion` of the rows (reference :245-265); rows
  [mh, H - mh] (inclusive) are kept."""
  _lib.require_cuda(pano)
  n, height, width, c = pano.shape
  masked_height = int(height * proportion)
  pano = pano.contiguous()
  out = torch.empty_like(pano)
  rc = _lib.lib().se3ds_mask_pano(_lib.ptr(pano), _lib.dtype_code(pano), n, height, width, c,
                                  masked_height, float(masked_region_value), _lib.ptr(out),
                                  _lib.stream())
  _lib.check(rc, 'se3ds_mask_pano')
  point_cloud_utils.propagate_int_range(out, pano, extra=(masked_region_value,))
  return out


def crop_pano(pano: torch.Tensor, proportion: float = 0.125, method: str = 'bilinear',
              resize_to_original: bool = False) -> torch.Tensor:
  """Removes the top and bottom `proportion` rows (reference :268-303); a view + copy."""
  if pano.dim() == 3:
    height = pano.shape[0]
  elif pano.dim() == 4:
    height = pano.shape[1]
  else:
    raise ValueError(f'pano should be of shape (N, H, W, C), got {tuple(pano.shape)} instead.')
  masked_height = int(height * proportion)
  if pano.dim() == 3:
    cropped = pano[masked_height:height - masked_height].contiguous()
  else:
    cropped = pano[:, masked_height:height - masked_height].contiguous()
  if resize_to_original:
    # reference :299-302: tf.image.resize(cropped, (H, W), method, antialias=True), cast back to
    # the input dtype.  The crop only removes rows, so the resize scales UP (rows x 1 / (1 - 2p),
    # columns x 1): TF's antialiasing only widens the kernel when scaling DOWN, so this is the
    # plain half-pixel-centre resize.
    if masked_height < 0 or 2 * masked_height >= height:
      raise ValueError('nothing left to resize')
    width = cropped.shape[-2]
    x4 = cropped[None] if cropped.dim() == 3 else cropped
    y = resize(x4, height, width, method)
    y = y[0] if cropped.dim() == 3 else y
    cropped = y if y.dtype == pano.dtype else y.to(pano.dtype)   # tf.cast: truncation for integers
  return cropped


def interpolate_bilinear(grid: torch.Tensor, query_points: torch.Tensor,
                         indexing: str = 'ij') -> torch.Tensor:
  """tfa.image.interpolate_bilinear semantics (floor clamped to [0, size-2], alpha clamped
  to [0, 1]).  grid (B,H,W,C) fp32, query (B,Q,2) -> (B,Q,C)."""
  if indexing not in ('ij', 'xy'):
    raise ValueError("Indexing mode must be 'ij' or 'xy'")
  _lib.require_cuda(grid, query_points)
  b, h, w, c = grid.shape
  if h < 2 or w < 2:
    raise ValueError('Grid must be at least 2x2.')
  q = query_points.shape[1]
  grid = grid.to(torch.float32).contiguous()
  query_points = query_points.to(torch.float32).contiguous()
  out = torch.empty((b, q, c), dtype=torch.float32, device=grid.device)
  rc = _lib.lib().se3ds_interp_bilinear(_lib.ptr(grid), _lib.ptr(query_points), b, h, w, c, q,
                                        1 if indexing == 'xy' else 0, _lib.ptr(out),
                                        _lib.stream())
  _lib.check(rc, 'se3ds_interp_bilinear')
  return out


def rotate_pano(pano: torch.Tensor, matrix: torch.Tensor,
                output_height: Optional[int] = None) -> torch.Tensor:
  """Rotates an equirect pano by (N,3,3) rotation matrices (reference :306-341)."""
  if pano.shape[2] != pano.shape[1] * 2:
    raise ValueError('Pano width must be twice height.')
  if output_height is not None:
    # reference :322-324 assigns into an immutable TensorShape and raises as well
    raise TypeError('output_height is not supported (the reference mutates a TensorShape)')
  _lib.require_cuda(pano, matrix)
  n, h, w, c = pano.shape
  rays = _host_tables.pixel_rays(h, pano.device)
  q = h * w
  matrix = matrix.to(torch.float32).contiguous()
  coords = torch.empty((n, q, 2), dtype=torch.float32, device=pano.device)
  rc = _lib.lib().se3ds_rotate_coords(_lib.ptr(rays), _lib.ptr(matrix), n, q, h, w,
                                      _lib.ptr(coords), _lib.stream())
  _lib.check(rc, 'se3ds_rotate_coords')
  return interpolate_bilinear(pano, coords).reshape(n, h, w, c)


def project_perspective_image(image, fov, output_height, camera_intrinsics=None, rotations=None,
                              rotation_matrix=None, pad_mode='constant', pad_value=0.0,
                              round_to_nearest=False):
  """Perspective (h,w,C) -> equirect (H,2H,C) (reference :344-417)."""
  assert pad_mode in {'reflect', 'constant', 'mean'}, ('Unsupported pad mode: %s' % pad_mode)
  _lib.require_cuda(image)
  image = image.to(torch.float32)
  ih, iw, c = image.shape
  w2i = get_world_to_image_transform((ih, iw), fov, camera_intrinsics=camera_intrinsics,
                                     rotations=rotations,
                                     rotation_matrix=rotation_matrix).to(image.device)
  rays = _host_tables.pixel_rays(output_height, image.device)
  q = output_height * 2 * output_height
  add = 0.0
  grid = image[None]
  if pad_mode != 'reflect':
    padded = torch.empty((1, ih + 2, iw + 2, c), dtype=torch.float32, device=image.device)
    if pad_mode == 'mean':
      # reference :403-407: constant_values = tf.reduce_mean(image); stays on the device
      image = image.contiguous()
      mean = torch.empty(1, dtype=torch.float32, device=image.device)
      _lib.check(_lib.lib().se3ds_mean_f32(_lib.ptr(image), image.numel(), _lib.ptr(mean),
                                            _lib.stream()), 'se3ds_mean_f32')
      padded.copy_(mean.expand_as(padded))
    else:
      padded.fill_(float(pad_value))
    padded[0, 1:-1, 1:-1] = image  # DtoD copy
    grid = padded
    add = 1.0
  coords = torch.empty((1, q, 2), dtype=torch.float32, device=image.device)
  rc = _lib.lib().se3ds_perspective_coords(_lib.ptr(rays), _lib.ptr(w2i), q,
                                           1 if round_to_nearest else 0, add, _lib.ptr(coords),
                                           _lib.stream())
  _lib.check(rc, 'se3ds_perspective_coords')
  out = interpolate_bilinear(grid, coords, indexing='xy')
  return out.reshape(output_height, 2 * output_height, c)


def get_perspective_from_equirectangular_image(image, camera_intrinsics, rotation_matrix, height,
                                               width):
  """Equirect (H,W,C) -> perspective (height,width,C) (reference :443-476)."""
  _lib.require_cuda(image)
  eq_height, eq_width, channels = image.shape
  k = np.asarray(camera_intrinsics.detach().cpu().numpy()
                 if isinstance(camera_intrinsics, torch.Tensor) else camera_intrinsics, np.float64)
  kinv_t = torch.from_numpy(np.ascontiguousarray(np.linalg.inv(k).astype(F32).T)).to(image.device)
  rot = torch.as_tensor(np.asarray(
      rotation_matrix.detach().cpu().numpy() if isinstance(rotation_matrix, torch.Tensor)
      else rotation_matrix, F32)).contiguous().to(image.device)
  uv = torch.empty((1, height * width, 2), dtype=torch.float32, device=image.device)
  rc = _lib.lib().se3ds_persp_from_equirect_coords(_lib.ptr(kinv_t), _lib.ptr(rot), height, width,
                                                   eq_height, eq_width, _lib.ptr(uv),
                                                   _lib.stream())
  _lib.check(rc, 'se3ds_persp_from_equirect_coords')
  out = interpolate_bilinear(image.to(torch.float32)[None], uv, indexing='xy')
  return out.reshape(height, width, channels)


def compact_valid_points(xyz1: torch.Tensor, feats: torch.Tensor, void_class: float):
  """Stream compaction of models.py:229-236: keep point j iff any(feats[:, j, :] != void).
  xyz1 (N,4,M), feats (N,M,C) -> (N,4,K), (N,K,C).  Synchronises to read K."""
  _lib.require_cuda(xyz1, feats)
  n, _, m = xyz1.shape
  c = feats.shape[-1]
  dev = xyz1.device
  if m == 0:
    return xyz1, feats
  xyz1 = xyz1.contiguous()
  feats = feats.contiguous()
  L = _lib.lib()
  ws = torch.empty((L.se3ds_compact_workspace_bytes(m),), dtype=torch.uint8, device=dev)
  xo = torch.empty_like(xyz1)
  fo = torch.empty_like(feats)
  cnt = torch.zeros((1,), dtype=torch.int64, device=dev)
  rc = L.se3ds_compact_valid(_lib.ptr(xyz1), _lib.ptr(feats), _lib.dtype_code(feats), n, m, c,
                             float(void_class), _lib.ptr(xo), _lib.ptr(fo), _lib.ptr(cnt),
                             _lib.ptr(ws), ws.numel(), _lib.stream())
  _lib.check(rc, 'se3ds_compact_valid')
  k = int(cnt.item())
  fk = fo[:, :k].contiguous()
  point_cloud_utils.propagate_int_range(fk, feats)   # (a compaction keeps values)
  return xo[:, :, :k].contiguous(), fk


# ------------------------------------------------------------------ fused perspective paths
def _np32(x):
  return np.asarray(x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x, F32)


def perspective_to_pointcloud(image: torch.Tensor, depth: torch.Tensor, output_height: int,
                              void_class: float, depth_scale: float, fov=None,
                              camera_intrinsics=None, rotations=None, rotation_matrix=None,
                              pad_value: float = 0.0, round_to_nearest: bool = True,
                              position: Optional[torch.Tensor] = None
                              ) -> Tuple[torch.Tensor, torch.Tensor]:
  """RE10K notebook cell 15 in one kernel: project_perspective_image(image) and
  (depth) (reference :344-417, constant padding), `tf.cast(rgb * 255, tf.int32)`, and
  equirectangular_to_pointcloud (:164-242) -- the 1024x2048 equirect RGB / depth intermediates are
  never written.  image (h,w,C) fp32 in [0,1], depth (h,w) fp32 in [0,1].  Returns xyz1
  (1,4,H*2H) fp32 and feats (1,H*2H,C) int32, bit-identical to the op-by-op chain."""
  _lib.require_cuda(image, depth, position)
  image = image.to(torch.float32).contiguous()
  depth = depth.to(torch.float32).contiguous()
  ih, iw, c = image.shape
  if tuple(depth.shape) != (ih, iw):
    raise ValueError(f'depth {tuple(depth.shape)} does not match the image {(ih, iw)}')
  dev = image.device
  h, w = output_height, 2 * output_height
  w2i = get_world_to_image_transform((ih, iw), fov, camera_intrinsics=camera_intrinsics,
                                     rotations=rotations, rotation_matrix=rotation_matrix).to(dev)
  rays = _host_tables.pixel_rays(h, dev)
  tab = _host_tables.equirect_tables(h, w, dev)
  base = tab.data_ptr()
  xyz1 = torch.empty((1, 4, h * w), dtype=torch.float32, device=dev)
  feats = torch.empty((1, h * w, c), dtype=torch.int32, device=dev)
  if position is not None:
    position = position.to(torch.float32).contiguous()
  rc = _lib.lib().se3ds_perspective_to_pointcloud(
      _lib.ptr(image), _lib.ptr(depth), ih, iw, c, _lib.ptr(rays), _lib.ptr(w2i),
      1 if round_to_nearest else 0, float(pad_value), base, base + 4 * h, base + 8 * h,
      base + 8 * h + 4 * w, _lib.ptr(position), h, w, float(void_class), float(depth_scale),
      _lib.ptr(xyz1), _lib.ptr(feats), _lib.stream())
  _lib.check(rc, 'se3ds_perspective_to_pointcloud')
  return xyz1, feats


def perspective_guidance(pred_rgb: torch.Tensor, pred_depth: torch.Tensor, camera_intrinsics,
                         rotation_matrix, height: int, width: int):
  """RE10K notebook cell 17 after the splat, in one kernel: the three
  get_perspective_from_equirectangular_image gathers (reference :443-476) of the projected RGB
  (H,W,3), depth (H,W) and validity mask (depth != 0, != 1, all(rgb != 0)), `rgb / 255` clipped to
  [0,1], `mask == 1`, and the products with the mask.  Returns the generator's proj_image
  (1,h,w,3), proj_depth (1,h,w,1), proj_mask (1,h,w,1)."""
  _lib.require_cuda(pred_rgb, pred_depth)
  pred_rgb = pred_rgb.to(torch.float32).contiguous()
  pred_depth = pred_depth.to(torch.float32).contiguous()
  eq_h, eq_w, c = pred_rgb.shape
  if c != 3 or tuple(pred_depth.shape) != (eq_h, eq_w):
    raise ValueError('pred_rgb (H,W,3) and pred_depth (H,W) expected')
  dev = pred_rgb.device
  k = np.asarray(_np32(camera_intrinsics), np.float64)
  kinv_t = torch.from_numpy(np.ascontiguousarray(np.linalg.inv(k).astype(F32).T)).to(dev)
  rot = torch.from_numpy(np.ascontiguousarray(_np32(rotation_matrix))).to(dev)
  f = lambda ch: torch.empty((1, height, width, ch), dtype=torch.float32, device=dev)
  proj_image, proj_depth, proj_mask = f(3), f(1), f(1)
  rc = _lib.lib().se3ds_perspective_guidance(
      _lib.ptr(pred_rgb), _lib.ptr(pred_depth), eq_h, eq_w, _lib.ptr(kinv_t), _lib.ptr(rot), height,
      width, _lib.ptr(proj_image), _lib.ptr(proj_depth), _lib.ptr(proj_mask), _lib.stream())
  _lib.check(rc, 'se3ds_perspective_guidance')
  return proj_image, proj_depth, proj_mask
