"""Host-side construction of the small fp32 tables the warp kernels consume (angle grids,
sin/cos tables, pixel rays, intrinsics).  These are O(H + W) values: they are built once per
shape in NumPy with TF's linspace semantics and cached on the device."""
import functools
import math

import numpy as np
import torch

F32 = np.float32


def linspace_f32(start, stop, num):
  """tf.linspace with Python-float endpoints: fp32; endpoints exact; interior points are
  start + delta * i with delta = (stop - start) / (num - 1), each op rounded to fp32."""
  start, stop, num = F32(start), F32(stop), int(num)
  if num == 1:
    return np.array([start], F32)
  delta = F32(F32(stop - start) / F32(num - 1))
  inner = (start + delta * np.arange(1, num - 1, dtype=F32)).astype(F32)
  return np.concatenate([[start], inner, [stop]]).astype(F32)


def sin32(a):
  """fp32 sine as the correctly rounded value of the fp64 sine of the fp32 angle."""
  return np.sin(np.asarray(a, np.float64)).astype(F32)


def cos32(a):
  return np.cos(np.asarray(a, np.float64)).astype(F32)


@functools.lru_cache(maxsize=64)
def _equirect_tables_np(height, width):
  # reference utils/pano_utils.py:211-218
  hp = 0.5 * np.pi / height
  elevation = linspace_f32(hp, np.pi - hp, height)
  heading = linspace_f32(1.5 * np.pi - hp, -0.5 * np.pi + hp, width)
  return np.concatenate([sin32(elevation), cos32(elevation), sin32(heading), cos32(heading)])


_dev_cache = {}


def _cached(key, device, make):
  k = (key, str(device))
  t = _dev_cache.get(k)
  if t is None:
    t = torch.from_numpy(np.ascontiguousarray(make())).to(device)
    _dev_cache[k] = t
  return t


def equirect_tables(height, width, device):
  """Device tensor [sin_el(H) | cos_el(H) | sin_hd(W) | cos_hd(W)]."""
  return _cached(('eqt', height, width), device, lambda: _equirect_tables_np(height, width))


def pixel_rays_np(output_height):
  """equirectangular_pixel_rays, reference utils/pano_utils.py:92-114 -> (3, H*W) fp32."""
  h = int(output_height)
  w = int(F32(h) * 2)
  heading = linspace_f32(-math.pi, math.pi, w)
  pitch = linspace_f32(0.0, math.pi, h)
  hh, pp = np.meshgrid(heading, pitch)
  xs = sin32(pp) * sin32(hh)
  ys = -cos32(pp)
  zs = sin32(pp) * cos32(hh)
  return np.stack([xs, ys, zs], 0).reshape(3, -1).astype(F32)


def pixel_rays(output_height, device):
  return _cached(('rays', int(output_height)), device, lambda: pixel_rays_np(output_height))


def perspective_grids(height, width, device):
  """linspace(-1, 1) grids of get_filtered_coords_and_feats (point_cloud_utils.py:59-62):
  computed in fp64 then cast to fp32."""
  return _cached(('pgrid', height, width), device, lambda: np.concatenate(
      [np.linspace(-1.0, 1.0, width).astype(F32), np.linspace(-1.0, 1.0, height).astype(F32)]))


def intrinsic_matrix_np(hfov):
  """get_intrinsic_matrix, point_cloud_utils.py:23-29."""
  f = 1 / np.tan(hfov / 2.)
  return np.array([[f, 0., 0., 0.], [0., f, 0., 0.], [0., 0., 1, 0], [0., 0., 0, 1]], F32)


def inv_intrinsics(hfov, device):
  return _cached(('kinv', float(hfov)), device,
                 lambda: np.linalg.inv(intrinsic_matrix_np(hfov).astype(np.float64)).astype(F32))
