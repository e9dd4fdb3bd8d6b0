"""CPU ORACLE (test infrastructure, NOT product code) -- NumPy restatement of the SE3DS
point-cloud warp: utils/pano_utils.py and utils/point_cloud_utils.py of the reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The shipped path (se3ds_amd/) never does; it fails loudly if the HIP library is missing.

Parity status: PINNED by the reference's own known-answer tests
  * utils/pano_utils_test.py:39-64   equirectangular_pixel_rays(3) golden array
  * models/models_test.py:81-137     plane-at-1m construction (axis values of the memory)
  * models/models_test.py:62-68      unproject -> project round trip (>= 95 % equal)
  * utils/pano_utils_test.py:67-136, utils/point_cloud_utils_test.py:26-64 shape/range
(tests/test_oracle_golden.py).  NOT pinned by any reference test (TF/tfa are absent here,
SURVEY.md section 8c): rotate_pano / project_perspective_image /
get_perspective_from_equirectangular_image and tfa.image.interpolate_bilinear; those follow
the published tfa 0.16.1 algorithm and are labelled "parity unpinned".

All arithmetic is fp32, one rounding per reference op, in the reference's order.
Transcendentals are evaluated in fp64 libm and rounded once to fp32 (= correctly rounded
fp32 whp); this is independent of include/se3ds_geom_math.h, which tests compare against it.
"""
import math

import numpy as np

F32 = np.float32
PI32 = F32(math.pi)
TWO_PI32 = F32(2 * math.pi)
ONE_HALF_PI32 = F32(1.5 * math.pi)
HFOV = 90 * 3.1415926535897932384626433 / 180  # constants.py:24-25


# ----------------------------------------------------------------------------- helpers
def linspace_f32(start, stop, num):
  """tf.linspace for Python-float endpoints (fp32): endpoints exact, interior
  start + delta * i with delta = (stop - start) / (num - 1), every op rounded to fp32."""
  start = F32(start)
  stop = F32(stop)
  num = int(num)
  if num == 1:
    return np.array([start], dtype=F32)
  delta = F32(F32(stop - start) / F32(num - 1))
  idx = np.arange(1, num - 1, dtype=F32)
  inner = (start + delta * idx).astype(F32)
  return np.concatenate([[start], inner, [stop]]).astype(F32)


def _sin32(a):
  return np.sin(np.asarray(a, dtype=np.float64)).astype(F32)


def _cos32(a):
  return np.cos(np.asarray(a, dtype=np.float64)).astype(F32)


def _atan2_32(y, x):
  return np.arctan2(np.asarray(y, np.float64), np.asarray(x, np.float64)).astype(F32)


def _acos32(w):
  with np.errstate(invalid='ignore'):
    return np.arccos(np.asarray(w, np.float64)).astype(F32)


def _asin32(w):
  with np.errstate(invalid='ignore'):
    return np.arcsin(np.asarray(w, np.float64)).astype(F32)


def _div_no_nan(a, b):
  a = np.asarray(a, F32)
  b = np.asarray(b, F32)
  with np.errstate(divide='ignore', invalid='ignore'):
    q = (a / b).astype(F32)
  return np.where(b == 0, F32(0), q).astype(F32)


def _resize_nearest(x, oh, ow):
  """tf.image.resize(method='nearest'), half-pixel centres: src = floor((i + 0.5) * scale)."""
  _, h, w, _ = x.shape
  if (oh, ow) == (h, w):
    return x
  sy = F32(h) / F32(oh)
  sx = F32(w) / F32(ow)
  iy = np.minimum(np.floor((np.arange(oh, dtype=F32) + F32(0.5)) * sy).astype(np.int64), h - 1)
  ix = np.minimum(np.floor((np.arange(ow, dtype=F32) + F32(0.5)) * sx).astype(np.int64), w - 1)
  return x[:, iy][:, :, ix]


def _resize_bilinear(x, oh, ow):
  """tf.image.resize(method='bilinear', antialias=False), half-pixel centres; fp32 out."""
  _, h, w, _ = x.shape
  if (oh, ow) == (h, w):
    return x.astype(F32)
  def coords(o, i):
    scale = F32(i) / F32(o)
    src = (np.arange(o, dtype=F32) + F32(0.5)) * scale - F32(0.5)
    lo = np.floor(src)
    frac = (src - lo).astype(F32)
    lo_i = np.clip(lo.astype(np.int64), 0, i - 1)
    hi_i = np.clip(lo.astype(np.int64) + 1, 0, i - 1)
    return lo_i, hi_i, frac
  y0, y1, fy = coords(oh, h)
  x0, x1, fx = coords(ow, w)
  xf = x.astype(F32)
  top = xf[:, y0][:, :, x0] + (xf[:, y0][:, :, x1] - xf[:, y0][:, :, x0]) * fx[None, None, :, None]
  bot = xf[:, y1][:, :, x0] + (xf[:, y1][:, :, x1] - xf[:, y1][:, :, x0]) * fx[None, None, :, None]
  return (top + (bot - top) * fy[None, :, None, None]).astype(F32)


# ------------------------------------------------------------------ point_cloud_utils.py
def get_intrinsic_matrix(hfov):
  """point_cloud_utils.py:23-29."""
  f = 1 / np.tan(hfov / 2.)
  return np.array([[f, 0, 0, 0], [0, f, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=F32)


def get_filtered_coords_and_feats(feats, depth, depth_scale):
  """point_cloud_utils.py:32-87 (perspective unproject; feats are int32 in the reference)."""
  feats = np.asarray(feats)
  if feats.ndim not in (3, 4):
    raise ValueError('feats should have shape (N, H, W) or (N, H, W, C),'
                     f' got {feats.shape} instead.')
  scalar = feats.ndim == 3
  if scalar:
    feats = feats[..., None]
  n, h, w = depth.shape
  c = feats.shape[-1]
  # tf.linspace(-1, 1, W) with Python ints yields float64 in TF 2.8; values are then cast
  # to fp32 (point_cloud_utils.py:59-62).
  xs = np.linspace(-1.0, 1.0, w).astype(F32)[None, None, None, :]
  ys = np.linspace(-1.0, 1.0, h).astype(F32)[None, None, :, None]
  d = (depth.astype(F32) * F32(depth_scale))[:, None]
  xyz = np.concatenate([np.broadcast_to(xs, d.shape) * d, np.broadcast_to(ys, d.shape) * d, d,
                        np.ones_like(d)], axis=1).astype(F32)
  dflat = d.reshape(n, -1)
  mask = (dflat > 0) & (dflat < F32(depth_scale))
  ff = feats.reshape(n, -1, c) * mask[..., None].astype(np.int32)
  ff = ff.astype(F32)
  xyz = xyz.reshape(n, 4, -1) * mask[:, None, :].astype(F32)
  kinv = np.linalg.inv(get_intrinsic_matrix(HFOV).astype(np.float64)).astype(F32)
  xyz = np.einsum('ij,njm->nim', kinv, xyz).astype(F32)
  if scalar:
    ff = ff[..., 0]
  return xyz, ff


def splat_indices(coords, feats, height, width, input_void_class):
  """Index half of project_to_feat (point_cloud_utils.py:124-152): per point the flat
  index (b*H*W + v*W + u) * valid, the valid flag and z.  Exposed for index-level tests."""
  x, y, z = coords[:, 0], coords[:, 1], coords[:, 2]
  vx = _div_no_nan(x, z)
  vy = _div_no_nan(y, z)
  fx = ((vx + F32(1)) / F32(2) * F32(width)).astype(F32)
  fy = ((vy + F32(1)) / F32(2) * F32(height)).astype(F32)
  with np.errstate(invalid='ignore'):
    # x86 float->int32: out of range / NaN -> INT_MIN
    def to_i32(f):
      ok = np.isfinite(f) & (f > -2147483904.0) & (f < 2147483648.0)
      return np.where(ok, np.trunc(np.where(ok, f, 0)), -2147483648.0).astype(np.int64)
    u = to_i32(fx)
    v = to_i32(fy)
  valid = (u >= 0) & (u < width) & (v >= 0) & (v < height)
  valid &= z > 0
  valid &= np.all(feats != F32(input_void_class), axis=-1)
  n = coords.shape[0]
  off = (np.arange(n, dtype=np.int64) * width * height)[:, None]
  flat = (off + v * width + u) * valid
  return flat.astype(np.int64), valid, z.astype(F32)


def project_to_feat(transformed_coords, feats, height, width, depth_scale, input_void_class,
                    output_void_class=0):
  """point_cloud_utils.py:90-183: z-buffer splat.  scatter-min depth, 0.1 m tolerance,
  per-channel scatter-max of the surviving features; invalid / culled points go to flat
  index 0 (batch 0, pixel (0,0)) and still take part in both scatters."""
  feats = np.asarray(feats)
  if feats.ndim not in (2, 3):
    raise ValueError('feats should have shape (N, M) or (N, M, C), got'
                     f' {feats.shape} instead.')
  scalar = feats.ndim == 2
  if scalar:
    feats = feats[..., None]
  coords = np.asarray(transformed_coords, F32)
  feats = feats.astype(F32)
  n = coords.shape[0]
  c = feats.shape[-1]
  flat, _, z = splat_indices(coords, feats, height, width, input_void_class)
  flat = flat.reshape(-1)
  zf = z.reshape(-1)
  zmin = np.full((n * height * width,), F32(depth_scale), dtype=F32)
  np.minimum.at(zmin, flat, zf)
  depth = (np.clip(zmin, F32(0), F32(depth_scale)) / F32(depth_scale)).astype(F32)
  depth = depth.reshape(n, height, width)
  keep = zf < (zmin[flat] + F32(0.1)).astype(F32)
  flat2 = flat * keep
  out = np.full((n * height * width, c), F32(output_void_class), dtype=F32)
  ff = feats.reshape(-1, c)
  for ch in range(c):
    np.maximum.at(out[:, ch], flat2, ff[:, ch])
  out = out.reshape(n, height, width, c)
  if scalar:
    out = out[..., 0]
  return depth, out


# ------------------------------------------------------------------------- pano_utils.py
def equirectangular_pixel_rays(output_height):
  """pano_utils.py:92-114.  Returns (3, H*W) fp32."""
  h = int(output_height)
  w = int(F32(h) * 2)
  heading = linspace_f32(-math.pi, math.pi, w)
  pitch = linspace_f32(0.0, math.pi, h)
  hh, pp = np.meshgrid(heading, pitch)
  xs = _sin32(pp) * _sin32(hh)
  ys = -_cos32(pp)
  zs = _sin32(pp) * _cos32(hh)
  return np.stack([xs, ys, zs], axis=0).reshape(3, -1).astype(F32)


def equirect_project_coords(xyz1):
  """pano_utils.py:139-156: world xyz -> (proj_x, proj_y, proj_z, 1), fp32 op for op."""
  xyz1 = np.asarray(xyz1, F32)
  x, y, z = xyz1[:, 0], xyz1[:, 1], xyz1[:, 2]
  rad = np.sqrt(((x * x + y * y) + z * z).astype(F32)).astype(F32)
  heading = _atan2_32(y, x)
  heading = (ONE_HALF_PI32 - heading).astype(F32)
  heading = (heading + TWO_PI32 * (heading <= 0).astype(F32)).astype(F32)
  heading = (heading - TWO_PI32 * (heading > TWO_PI32).astype(F32)).astype(F32)
  elevation = _acos32(_div_no_nan(z, rad))
  px = (rad * ((heading / TWO_PI32) * F32(2) - F32(1))).astype(F32)
  py = (rad * ((elevation / PI32) * F32(2) - F32(1))).astype(F32)
  return np.stack([px, py, rad, np.ones_like(px)], axis=1).astype(F32)


def project_feats_to_equirectangular(feats, xyz1, height, width, void_class, depth_scale):
  """pano_utils.py:117-161."""
  proj = equirect_project_coords(xyz1)
  return project_to_feat(proj, np.asarray(feats).astype(F32), height, width,
                         depth_scale=depth_scale, input_void_class=void_class)


def equirect_angle_tables(height, width):
  """sin/cos tables of the half-pixel-centre elevation / heading grids
  (pano_utils.py:211-218): returns sin_el (H), cos_el (H), sin_hd (W), cos_hd (W)."""
  hp = 0.5 * np.pi / height
  elevation = linspace_f32(hp, np.pi - hp, height)
  heading = linspace_f32(1.5 * np.pi - hp, -0.5 * np.pi + hp, width)
  return _sin32(elevation), _cos32(elevation), _sin32(heading), _cos32(heading)


def equirectangular_to_pointcloud(feats, depth, void_class, depth_scale, size_mult=1.0,
                                  interpolation_method='nearest'):
  """pano_utils.py:164-242."""
  feats = np.asarray(feats)
  if feats.ndim not in (3, 4):
    raise ValueError('feats should have shape (N, H, W) or (N, H, W, C),'
                     f' got {feats.shape} instead.')
  if void_class < 0.0 and feats.dtype in (np.uint8, np.uint16, np.uint32, np.uint64):
    raise ValueError('feats datatype must be signed if the void class is negative')
  scalar = feats.ndim == 3
  if scalar:
    feats = feats[..., None]
  n, h, w, c = feats.shape
  assert w == 2 * h, 'Expected equirectangular input images'
  sh, sw = int(h * size_mult), int(w * size_mult)
  depth = np.asarray(depth, F32)
  pano_depth = _resize_nearest(depth[..., None], sh, sw)[..., 0]
  if interpolation_method == 'nearest':
    pano_feats = _resize_nearest(feats, sh, sw)
  elif (sh, sw) == (h, w):
    pano_feats = feats.astype(F32)  # bilinear tf.image.resize always returns fp32
  else:
    pano_feats = _resize_bilinear(feats, sh, sw)
  sin_el, cos_el, sin_hd, cos_hd = equirect_angle_tables(sh, sw)
  mask = ((pano_depth > 0) & (pano_depth < F32(1.0))).astype(F32)
  rad = ((pano_depth * F32(depth_scale)).astype(F32) * mask).astype(F32)
  pano_feats = np.where(mask[..., None] == 0, np.asarray(void_class).astype(pano_feats.dtype),
                        pano_feats)
  x = ((rad * sin_el[None, :, None]).astype(F32) * cos_hd[None, None, :]).astype(F32)
  y = ((rad * sin_el[None, :, None]).astype(F32) * sin_hd[None, None, :]).astype(F32)
  z = (rad * cos_el[None, :, None]).astype(F32)
  xyz1 = np.stack([x.reshape(n, -1), y.reshape(n, -1), z.reshape(n, -1),
                   np.ones((n, sh * sw), F32)], axis=1)
  out = pano_feats.reshape(n, -1, c)
  if scalar:
    out = out[..., 0]
  return xyz1, out


def mask_pano(pano, proportion=0.125, masked_region_value=0):
  """pano_utils.py:245-265 (rows mh .. H-mh inclusive are kept)."""
  pano = np.asarray(pano)
  h = pano.shape[1]
  mh = int(h * proportion)
  r = np.arange(h)
  m = ((r >= mh) & (r <= h - mh)).astype(pano.dtype)[None, :, None, None]
  return m * pano + (1 - m) * np.asarray(masked_region_value).astype(pano.dtype)


def crop_pano(pano, proportion=0.125, method='bilinear', resize_to_original=False):
  """pano_utils.py:268-303 (rows mh .. H-mh-1).  resize_to_original: tf.image.resize(...,
  antialias=True) back to (H, W) and tf.cast to the input dtype; the crop only removes rows, so
  the resize scales UP, where TF's antialiasing (a kernel widened by max(1 / scale, 1)) changes
  nothing -- PARITY UNPINNED (no reference test touches this branch)."""
  pano = np.asarray(pano)
  if pano.ndim not in (3, 4):
    raise ValueError(f'pano should be of shape (N, H, W, C), got {pano.shape} instead.')
  x = pano[None] if pano.ndim == 3 else pano
  h, w = x.shape[1], x.shape[2]
  mh = int(h * proportion)
  x = x[:, mh:h - mh]
  if resize_to_original:
    x = _resize_nearest(x, h, w) if method == 'nearest' else _resize_bilinear(x, h, w)
    x = np.trunc(x).astype(pano.dtype) if np.issubdtype(pano.dtype, np.integer) else x.astype(pano.dtype)
  return x[0] if pano.ndim == 3 else x


def interpolate_bilinear(grid, query_points, indexing='ij'):
  """tfa.image.interpolate_bilinear (tensorflow-addons 0.16.1, dense_image_warp.py).
  grid (B,H,W,C) fp32, query (B,Q,2).  parity unpinned by the reference's tests."""
  grid = np.asarray(grid, F32)
  q = np.asarray(query_points, F32)
  b, h, w, c = grid.shape
  order = [0, 1] if indexing == 'ij' else [1, 0]
  floors, ceils, alphas = [], [], []
  for i, dim in enumerate(order):
    queries = q[..., dim]
    size = (h, w)[i]
    max_floor = F32(size - 2)
    fl = np.minimum(np.maximum(F32(0), np.floor(queries)), max_floor)
    fi = fl.astype(np.int64)
    floors.append(fi)
    ceils.append(fi + 1)
    alpha = (queries - fl).astype(F32)
    alpha = np.minimum(np.maximum(F32(0), alpha), F32(1))
    alphas.append(alpha[..., None])
  flat = grid.reshape(b * h * w, c)
  off = (np.arange(b) * h * w)[:, None]
  def gather(yc, xc):
    return flat[off + yc * w + xc]
  tl = gather(floors[0], floors[1])
  tr = gather(floors[0], ceils[1])
  bl = gather(ceils[0], floors[1])
  br = gather(ceils[0], ceils[1])
  top = (alphas[1] * (tr - tl) + tl).astype(F32)
  bot = (alphas[1] * (br - bl) + bl).astype(F32)
  return (alphas[0] * (bot - top) + top).astype(F32)


def rotate_coords(matrix, height, width_src, height_src):
  """Coordinate half of rotate_pano (pano_utils.py:326-338): (N, Q, 2) [pitch_px, heading_px]."""
  rays = equirectangular_pixel_rays(height)  # (3, Q)
  m = np.asarray(matrix, F32)
  rot = np.matmul(m, rays[None]).astype(F32)  # fp32 matmul, k = 3
  x, y, z = rot[:, 0], rot[:, 1], rot[:, 2]
  pitch = _acos32(-y)
  heading = _atan2_32(x, z)
  hp = ((heading / TWO_PI32 + F32(0.5)) * F32(width_src - 1)).astype(F32)
  pp = (pitch / PI32 * F32(height_src - 1)).astype(F32)
  return np.stack([pp, hp], axis=-1)


def rotate_pano(pano, matrix):
  """pano_utils.py:306-341 (output_height=None; the resize branch mutates a TensorShape and
  cannot run in the reference)."""
  pano = np.asarray(pano, F32)
  n, h, w, c = pano.shape
  if w != h * 2:
    raise ValueError('Pano width must be twice height.')
  coords = rotate_coords(matrix, h, w, h)
  return interpolate_bilinear(pano, coords).reshape(n, h, w, c)


def get_world_to_image_transform(image_shape, fov, camera_intrinsics=None, rotations=None,
                                 rotation_matrix=None):
  """pano_utils.py:26-89 (fp32)."""
  if camera_intrinsics is None:
    height, width = F32(image_shape[0]), F32(image_shape[1])
    fov_y, fov_x = F32(fov[0]), F32(fov[1])
    tan32 = lambda a: F32(np.tan(np.float64(a)))
    fx = F32(F32(0.5) * (width - F32(1.0))) / tan32(fov_x / F32(2))
    fy = F32(F32(0.5) * (height - F32(1.0))) / tan32(fov_y / F32(2))
    camera_intrinsics = np.array([[fx, 0, F32(0.5) * (width - F32(1))],
                                  [0, fy, F32(0.5) * (height - F32(1))], [0, 0, 1]], F32)
  camera_intrinsics = np.asarray(camera_intrinsics, F32)
  if rotations is not None:
    rp, rh = F32(rotations[0]), F32(rotations[1])
    pitch_rot = np.array([[1, 0, 0], [0, _cos32(-rp), -_sin32(-rp)],
                          [0, _sin32(-rp), _cos32(-rp)]], F32)
    head_rot = np.array([[_cos32(-rh), 0, _sin32(-rh)], [0, 1, 0],
                         [-_sin32(-rh), 0, _cos32(-rh)]], F32)
    extr = np.matmul(pitch_rot, head_rot).astype(F32)
  elif rotation_matrix is not None:
    extr = np.asarray(rotation_matrix, F32)
  else:
    extr = np.eye(3, dtype=F32)
  return np.matmul(camera_intrinsics, extr).astype(F32)


def perspective_coords(world_to_image, output_height, round_to_nearest=False):
  """Coordinate half of project_perspective_image (pano_utils.py:387-402), before padding."""
  rays = equirectangular_pixel_rays(output_height)
  ic = np.matmul(np.asarray(world_to_image, F32), rays).astype(F32).T  # (Q, 3)
  xy = ic[:, :2]
  zs = ic[:, 2:]
  with np.errstate(divide='ignore', invalid='ignore'):
    coords = np.where(zs > 0, (xy / zs).astype(F32), F32(-1))
  if round_to_nearest:
    coords = np.round(coords).astype(F32)  # tf.math.round = half to even
  return coords.astype(F32)


def project_perspective_image(image, fov, output_height, camera_intrinsics=None, rotations=None,
                              rotation_matrix=None, pad_mode='constant', pad_value=0.0,
                              round_to_nearest=False):
  """pano_utils.py:344-417.  image (h, w, C) -> (H, 2H, C)."""
  assert pad_mode in {'reflect', 'constant', 'mean'}, ('Unsupported pad mode: %s' % pad_mode)
  image = np.asarray(image, F32)[None]
  w2i = get_world_to_image_transform((image.shape[1], image.shape[2]), fov,
                                     camera_intrinsics=camera_intrinsics, rotations=rotations,
                                     rotation_matrix=rotation_matrix)
  coords = perspective_coords(w2i, output_height, round_to_nearest)
  if pad_mode != 'reflect':
    cv = F32(np.mean(image.astype(np.float64))) if pad_mode == 'mean' else F32(pad_value)
    image = np.pad(image, ((0, 0), (1, 1), (1, 1), (0, 0)), mode='constant', constant_values=cv)
    coords = (coords + F32(1.0)).astype(F32)
  out = interpolate_bilinear(image, coords[None], indexing='xy')
  return out.reshape(output_height, 2 * output_height, -1)


def _xyz_to_lonlat(xyz):
  """pano_utils.py:420-433."""
  xyz = np.asarray(xyz, F32)
  norm = np.sqrt(np.sum((xyz * xyz).astype(F32), axis=-1, keepdims=True, dtype=F32)).astype(F32)
  xn = (xyz / norm).astype(F32)
  lon = _atan2_32(xn[..., 0:1], xn[..., 2:])
  lat = _asin32(xn[..., 1:2])
  return np.concatenate([lon, lat], axis=-1)


def _lonlat_to_uv(lonlat, shape):
  """pano_utils.py:436-440."""
  u = ((lonlat[..., 0:1] / TWO_PI32 + F32(0.5)) * F32(shape[1] - 1)).astype(F32)
  v = ((lonlat[..., 1:] / PI32 + F32(0.5)) * F32(shape[0] - 1)).astype(F32)
  return np.concatenate([u, v], axis=-1)


def perspective_from_equirect_coords(camera_intrinsics, rotation_matrix, height, width, eq_shape):
  """Coordinate half of get_perspective_from_equirectangular_image (pano_utils.py:459-469)."""
  x, y = np.meshgrid(np.arange(width), np.arange(height))
  xyz = np.stack([x, y, np.ones_like(x)], axis=-1).astype(F32)
  kinv_t = np.linalg.inv(np.asarray(camera_intrinsics, np.float64)).astype(F32).T
  xyz = np.matmul(np.matmul(xyz, kinv_t).astype(F32), np.asarray(rotation_matrix, F32)).astype(F32)
  uv = _lonlat_to_uv(_xyz_to_lonlat(xyz), eq_shape)
  return uv.reshape(-1, 2).astype(F32)


def get_perspective_from_equirectangular_image(image, camera_intrinsics, rotation_matrix, height,
                                               width):
  """pano_utils.py:443-476.  image (H, W, C) -> (height, width, C)."""
  image = np.asarray(image)
  eh, ew, c = image.shape
  uv = perspective_from_equirect_coords(camera_intrinsics, rotation_matrix, height, width,
                                        (eh, ew))
  out = interpolate_bilinear(image.astype(F32)[None], uv[None], indexing='xy')
  return out.reshape(height, width, c)


# ------------------------------------------------------------------------ models/models.py
def proj_mask(proj_depth, proj_rgb, void=-1):
  """models.py:282-287: (depth > 0) & (depth < 1) & all(rgb != void) -> (N,H,W,1) fp32."""
  m = (proj_depth > 0) & (proj_depth < 1) & np.all(proj_rgb != void, axis=-1)
  return m.astype(F32)[..., None]


def compact_valid(xyz1, feats, void):
  """models.py:229-236: keep point j if any(feats[:, j, :] != void) over batch and channel."""
  valid = np.any(feats != void, axis=(0, 2))
  idx = np.nonzero(valid)[0]
  return xyz1[:, :, idx], feats[:, idx]


# ------------------------------------------------- notebooks/SE3DS_RE10K_Colab.ipynb cells 15 / 17
def notebook_cell15_pointcloud(rgb01, depth01, camera_intrinsics, rotation_matrix, eq_height,
                               void_class=-1, depth_scale=20.0):
  """Cell 15: two project_perspective_image calls (round_to_nearest=True, constant padding),
  `tf.cast(rgb * 255, tf.int32)`, equirectangular_to_pointcloud.  rgb01 (h,w,3) fp32 in [0,1],
  depth01 (h,w) fp32 in [0,1] -> xyz1 (1,4,P), feats (1,P,3) int32."""
  rgb_t = project_perspective_image(rgb01, None, eq_height, camera_intrinsics=camera_intrinsics,
                                    rotation_matrix=rotation_matrix, round_to_nearest=True)
  depth_t = project_perspective_image(np.asarray(depth01, F32)[..., None], None, eq_height,
                                      camera_intrinsics=camera_intrinsics,
                                      rotation_matrix=rotation_matrix, round_to_nearest=True)
  proj_depth = depth_t[None, ..., 0]
  proj_rgb = (rgb_t[None] * F32(255)).astype(F32).astype(np.int32)   # tf.cast truncates
  return equirectangular_to_pointcloud(proj_rgb, proj_depth, void_class, depth_scale)


def notebook_cell17_guidance(pred_rgb, pred_depth, camera_intrinsics, new_rotation_matrix,
                             pers_height, pers_width):
  """Cell 17 after the splat: the three get_perspective_from_equirectangular_image gathers, the
  `/ 255` + clip, the validity mask (depth != 1, != 0, all(rgb != 0)) gathered and compared with
  1.0, and the products that form the generator's inputs.  pred_rgb (H,W,3), pred_depth (H,W)
  fp32 -> proj_image (1,h,w,3), proj_depth (1,h,w,1), proj_mask (1,h,w,1)."""
  g = lambda img: get_perspective_from_equirectangular_image(img, camera_intrinsics,
                                                             new_rotation_matrix, pers_height,
                                                             pers_width)
  rgb_g = np.clip((g(pred_rgb) / F32(255)).astype(F32), 0, 1)[None]
  depth_g = g(np.asarray(pred_depth, F32)[..., None])[None]
  m = ((pred_depth != 1.0) & (pred_depth != 0.0) & np.all(pred_rgb != 0.0, axis=-1)).astype(F32)
  mask_g = (g(m[..., None])[None] == 1.0).astype(F32)
  return (mask_g * rgb_g).astype(F32), (mask_g * depth_g).astype(F32), mask_g
