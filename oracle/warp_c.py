"""CPU ORACLE (test infrastructure, NOT product code) -- ctypes front-end of
oracle/_build/libwarp_oracle.so (oracle/warp_oracle.c, the plain-C restatement of
utils/pano_utils.py:139-156 and utils/point_cloud_utils.py:124-176).  Built by
`make -C oracle` / __graft_entry__.build().  Only tests/, smoke() and bench.py's
cpu_baseline leg may import this."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libwarp_oracle.so')
_lib = None
F32 = np.float32


def build():
  subprocess.check_call(['make', '-C', _HERE, '-s'])


def lib():
  global _lib
  if _lib is None:
    if not os.path.exists(_SO):
      build()
    _lib = ctypes.CDLL(_SO)
  return _lib


def _p(a):
  return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
  return None if a is None else np.ascontiguousarray(a, dtype=F32)


def atan2f(y, x):
  y, x = _f32(y), _f32(x)
  out = np.empty_like(y)
  lib().oracle_atan2f(_p(y), _p(x), _p(out), ctypes.c_int64(y.size))
  return out


def acosf(w):
  w = _f32(w)
  out = np.empty_like(w)
  lib().oracle_acosf(_p(w), _p(out), ctypes.c_int64(w.size))
  return out


def asinf(w):
  w = _f32(w)
  out = np.empty_like(w)
  lib().oracle_asinf(_p(w), _p(out), ctypes.c_int64(w.size))
  return out


def equirect_project_coords(xyz1, offset=None):
  xyz1 = _f32(xyz1)
  n, _, m = xyz1.shape
  offset = _f32(offset)
  out = np.empty_like(xyz1)
  lib().oracle_equirect_project(_p(xyz1), _p(offset), _p(out), ctypes.c_int(n), ctypes.c_int64(m))
  return out


def fast_screen_stats(xyz, height, width):
  """(max |dfx|/W, max |dfy|/H, decided, wrong) of the fast index screen on xyz (3,M)."""
  xyz = _f32(xyz)
  out = np.zeros(4, np.float64)
  lib().oracle_fast_screen_stats(_p(xyz), ctypes.c_int64(xyz.shape[1]), ctypes.c_int(width),
                                 ctypes.c_int(height), out.ctypes.data_as(ctypes.c_void_p))
  return out


def project_to_feat(coords, feats, height, width, depth_scale, input_void_class,
                    output_void_class=0, return_flat=False):
  coords = _f32(coords)
  feats = np.asarray(feats)
  scalar = feats.ndim == 2
  if scalar:
    feats = feats[..., None]
  feats = _f32(feats)
  n, _, m = coords.shape
  c = feats.shape[-1]
  depth = np.empty((n, height, width), F32)
  feat = np.empty((n, height, width, c), F32)
  flat = np.empty((n * m,), np.int64) if return_flat else None
  lib().oracle_project_to_feat(_p(coords), _p(feats), ctypes.c_int(n), ctypes.c_int64(m),
                               ctypes.c_int(c), ctypes.c_int(height), ctypes.c_int(width),
                               ctypes.c_float(depth_scale), ctypes.c_float(input_void_class),
                               ctypes.c_float(output_void_class), _p(depth), _p(feat), _p(flat))
  if scalar:
    feat = feat[..., 0]
  if return_flat:
    return depth, feat, flat
  return depth, feat


def project_feats_to_equirectangular(feats, xyz1, height, width, void_class, depth_scale,
                                     offset=None):
  xyz1 = _f32(xyz1)
  feats = np.asarray(feats)
  scalar = feats.ndim == 2
  if scalar:
    feats = feats[..., None]
  feats = _f32(feats)
  n, _, m = xyz1.shape
  c = feats.shape[-1]
  offset = _f32(offset)
  depth = np.empty((n, height, width), F32)
  feat = np.empty((n, height, width, c), F32)
  lib().oracle_project_feats_to_equirect(_p(xyz1), _p(offset), _p(feats), ctypes.c_int(n),
                                         ctypes.c_int64(m), ctypes.c_int(c), ctypes.c_int(height),
                                         ctypes.c_int(width), ctypes.c_float(depth_scale),
                                         ctypes.c_float(void_class), _p(depth), _p(feat))
  if scalar:
    feat = feat[..., 0]
  return depth, feat


def unproject_equirect(feats, depth, tables, void_class, depth_scale, position=None):
  """feats (N,H,W,C) any dtype (computed in fp32), depth (N,H,W)."""
  feats = _f32(feats)
  depth = _f32(depth)
  n, h, w, c = feats.shape
  sin_el, cos_el, sin_hd, cos_hd = [_f32(t) for t in tables]
  position = _f32(position)
  xyz1 = np.empty((n, 4, h * w), F32)
  fo = np.empty((n, h * w, c), F32)
  lib().oracle_unproject_equirect(_p(feats), _p(depth), _p(sin_el), _p(cos_el), _p(sin_hd),
                                  _p(cos_hd), _p(position), ctypes.c_int(n), ctypes.c_int(h),
                                  ctypes.c_int(w), ctypes.c_int(c), ctypes.c_float(void_class),
                                  ctypes.c_float(depth_scale), _p(xyz1), _p(fo))
  return xyz1, fo
