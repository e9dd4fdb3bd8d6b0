"""Utility functions for processing 3D point clouds -- MI355X implementation of the
reference's utils/point_cloud_utils.py (same names, arguments and error behaviour; torch
CUDA tensors in place of tf.Tensor).  All arithmetic runs in libse3ds_hip.so."""
import os
from typing import Optional, Tuple

import torch

from se3ds_amd import _lib
from se3ds_amd import constants
from se3ds_amd.utils import _host_tables


def get_intrinsic_matrix(hfov: float) -> torch.Tensor:
  """Returns the intrinsic for a given horizontal FOV (reference :23-29)."""
  return torch.from_numpy(_host_tables.intrinsic_matrix_np(hfov))


def get_filtered_coords_and_feats(feats: torch.Tensor, depth: torch.Tensor, depth_scale: float):
  """Perspective unprojection (reference :32-87).

  feats (N,H,W) or (N,H,W,C) int32, depth (N,H,W) in [0,1].
  Returns xyz (N,4,H*W) fp32 and filtered feats (N,H*W[,C]) fp32."""
  if feats.dim() != 3 and feats.dim() != 4:
    raise ValueError('feats should have shape (N, H, W) or (N, H, W, C),'
                     f' got {tuple(feats.shape)} instead.')
  _lib.require_cuda(feats, depth)
  is_scalar = feats.dim() == 3
  if is_scalar:
    feats = feats[..., None]
  n, h, w = depth.shape
  c = feats.shape[-1]
  feats = feats.to(torch.int32).contiguous()
  depth = depth.to(torch.float32).contiguous()
  grids = _host_tables.perspective_grids(h, w, depth.device)
  kinv = _host_tables.inv_intrinsics(constants.HFOV, depth.device)
  xyz = torch.empty((n, 4, h * w), dtype=torch.float32, device=depth.device)
  out = torch.empty((n, h * w, c), dtype=torch.float32, device=depth.device)
  rc = _lib.lib().se3ds_unproject_perspective(
      _lib.ptr(feats), _lib.ptr(depth), grids.data_ptr(), grids.data_ptr() + 4 * w,
      _lib.ptr(kinv), n, h, w, c, float(depth_scale), _lib.ptr(xyz), _lib.ptr(out), _lib.stream())
  _lib.check(rc, 'se3ds_unproject_perspective')
  if is_scalar:
    out = out[..., 0]
  return xyz, out


_workspaces = {}
_WS_BYTES = {}   # (n, m, height, width, channels) -> se3ds_splat_workspace_bytes


def _dev_key(device) -> str:
  """ONE key per physical device: 'cuda', torch.device('cuda') and 'cuda:0' name the same workspace
  (a memory created on torch.device('cuda') stored its workspace under 'cuda' while the
  promise poll looked under 'cuda:0', found nothing and skipped silently)."""
  d = torch.device(device)
  if d.type == 'cuda' and d.index is None:
    d = torch.device('cuda', torch.cuda.current_device())
  return str(d)


def _workspace(nbytes: int, device) -> torch.Tensor:
  """Grow-only scratch buffer per device (keeps allocation out of the trajectory loop).  Its
  256-byte header is zeroed at creation: header word 3 is the library's STICKY promise-violation
  flag (include/se3ds_hip.h, se3ds_splat_promise_sticky); words 1 and 2 are the single-launch
  splat's grid barrier and ticket (zero at rest)."""
  key = _dev_key(device)
  ws = _workspaces.get(key)
  if ws is None or ws.numel() < nbytes:
    _poll_promise(device, force=True)   # (a pending verdict of the old buffer is read first)
    ws = torch.empty((max(nbytes, 1 << 20),), dtype=torch.uint8, device=device)
    ws[:256].zero_()
    _workspaces[key] = ws
  return ws


FEAT_BYTE_RANGE = 0x100   # include/se3ds_hip.h SE3DS_FEAT_BYTE_RANGE
XYZ1_ONES_PRESET = 0x200  # include/se3ds_hip.h SE3DS_XYZ1_ONES_PRESET
# A broken byte-range promise makes the packed splat pack f & 255: silently wrong features.  The
# kernels raise a sticky flag in the workspace; it is read back ASYNCHRONOUSLY after the FIRST
# promised splat of a process and after every _PROMISE_POLL_EVERY-th (a 1-thread kernel + a 4-byte
# copy into pinned memory, examined at a later call without waiting).  Between polls a violation is
# only PENDING: check_promise(device) forces the read-back and raises -- the library's own sync
# points call it (SE3DSModel.__call__ on return, write_memory_as_pointcloud, generated_rollout's
# end) and an atexit hook reports a violation nobody polled for.  SE3DS_CHECK_PROMISE=1: checked
# synchronously after every promised splat (debugging).
_PROMISE_POLL_EVERY = 64
_CHECK_PROMISE_SYNC = os.environ.get('SE3DS_CHECK_PROMISE') == '1'
_promise_state = {}   # device -> dict(calls, pending=[(event, pinned flag)])


class PromiseBroken(RuntimeError):
  """An int32 feature outside [0, 255] (and != void) reached a splat that was promised
  SE3DS_FEAT_BYTE_RANGE: the features it rendered are wrong."""


def _poll_promise(device, force=False, launch=False):
  """Bookkeeping of the sticky flag (see above).  force: launch the read-back now and wait;
  launch: launch it now, examine it at a later call."""
  key = _dev_key(device)
  st = _promise_state.setdefault(key, dict(calls=0, pending=[]))
  ws = _workspaces.get(key)
  still = []
  for ev, flag in st['pending']:
    if force or ev.query():
      ev.synchronize()
      if int(flag[0]) != 0:
        st['pending'] = []
        raise PromiseBroken('SE3DS_FEAT_BYTE_RANGE was promised for int32 features that are not all '
                            'void or in [0, 255]: the splats rendered since the last check are wrong '
                            '(call point_cloud_utils.set_byte_range(t, None) after writing to a '
                            'feature tensor through raw pointers)')
    else:
      still.append((ev, flag))
  st['pending'] = still
  if ws is None:
    return
  st['calls'] += 1
  if (force or launch or _CHECK_PROMISE_SYNC or st['calls'] == 1 or
      st['calls'] % _PROMISE_POLL_EVERY == 0):
    dflag = torch.empty(1, dtype=torch.int32, device=ws.device)
    _lib.check(_lib.lib().se3ds_splat_promise_sticky(_lib.ptr(ws), _lib.ptr(dflag), 1, _lib.stream()),
               'se3ds_splat_promise_sticky')
    hflag = torch.empty(1, dtype=torch.int32, pin_memory=True)
    hflag.copy_(dflag, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    st['pending'].append((ev, hflag))
    if force or _CHECK_PROMISE_SYNC:
      _poll_promise_wait(st)


def check_promise(device=None, wait=True):
  """Poll of the sticky byte-range flag on `device` (default: every device that has a splat
  workspace): launches the read-back (one 1-thread kernel + a 4-byte copy) and, with wait=True,
  WAITS for it and raises PromiseBroken if an int32 feature outside [0, 255] reached a promised
  splat since the last check; wait=False examines the read-backs that have completed and leaves the
  new one for the next call (no host synchronisation: SE3DSModel.__call__ does this on every
  return, so a short inference trajectory that never reaches the 64th promised splat is covered
  one call late at most, and by the atexit hook for its last call)."""
  keys = [_dev_key(device)] if device is not None else list(_workspaces)
  for key in keys:
    if key in _workspaces:
      _poll_promise(key, force=wait, launch=not wait)


def _check_promise_at_exit():
  try:
    if torch.cuda.is_available():
      check_promise()
  except PromiseBroken as e:   # (an exception out of atexit is only printed: say it clearly)
    import sys
    sys.stderr.write(f'se3ds_amd: {e}\n')
  except Exception:   # noqa: BLE001  (interpreter teardown: the runtime may be gone)
    pass


import atexit   # noqa: E402
atexit.register(_check_promise_at_exit)


def _poll_promise_wait(st):
  pend, st['pending'] = st['pending'], []
  for ev, flag in pend:
    ev.synchronize()
    if int(flag[0]) != 0:
      raise PromiseBroken('SE3DS_FEAT_BYTE_RANGE was promised for int32 features that are not all '
                          'void or in [0, 255]: the last splat rendered wrong features')


def set_byte_range(t: torch.Tensor, ok: Optional[bool], void_class: float = None):
  """Records on the tensor whether its elements are all `void_class` or integers in [0, 255] (what
  the 8-byte packed splat needs from int32 features).  Writers that fill a tensor through raw
  pointers (se3ds_unproject_equirect_into) call this; None forgets.  The verdict is only reused
  for the SAME void class."""
  t._se3ds_byte_range = None if ok is None else (
      bool(ok), t.data_ptr(), t._version, tuple(t.shape), None if void_class is None else float(void_class))


def set_int_range(t: torch.Tensor, lo: int, hi: int):
  """Records that every element of the integer tensor `t` lies in [lo, hi] BY CONSTRUCTION (the
  library's own quantisation kernel clamps to these bounds): byte_range() then answers without
  reading the tensor back."""
  t._se3ds_int_range = (int(lo), int(hi), t.data_ptr(), t._version, tuple(t.shape))


def get_int_range(t: torch.Tensor):
  """(lo, hi) recorded by set_int_range while the tensor is unchanged, else None."""
  rng = getattr(t, '_se3ds_int_range', None)
  if rng is not None and rng[2:] == (t.data_ptr(), t._version, tuple(t.shape)):
    return rng[0], rng[1]
  return None


def propagate_int_range(dst: torch.Tensor, *srcs, extra=()):
  """dst holds only values of `srcs` (copies, gathers, concatenations, compactions) and the
  integers in `extra` (a void class): its bounds are the union of theirs -- when every source's
  bounds are known.  Keeps byte_range() free of device read-backs along the model's call graph
  (quantise -> mask_pano -> unproject -> compact -> concat -> splat)."""
  if dst.dtype != torch.int32:
    return
  lo = hi = None
  for t in srcs:
    if t.numel() == 0:
      continue
    r = get_int_range(t)
    if r is None:
      return
    lo = r[0] if lo is None else min(lo, r[0])
    hi = r[1] if hi is None else max(hi, r[1])
  for v in extra:
    if float(v) != int(v):
      return
    lo = int(v) if lo is None else min(lo, int(v))
    hi = int(v) if hi is None else max(hi, int(v))
  if lo is not None:
    set_int_range(dst, lo, hi)


def byte_range(t: torch.Tensor, void_class: float) -> bool:
  """True when the splat may be promised SE3DS_FEAT_BYTE_RANGE for `t` (uint8: by type; int32:
  every element is `void_class` or in [0, 255]).  Known bounds (set_int_range) answer at once;
  otherwise the int32 answer is computed ONCE per tensor and void class (`se3ds_feats_byte_range`
  + one host read) and cached on the tensor object, keyed by its storage pointer, shape, torch
  version counter and the void class; the library's own in-place writers keep the cache current."""
  if t.dtype == torch.uint8:
    return True
  if t.dtype != torch.int32:
    return False
  same = (t.data_ptr(), t._version, tuple(t.shape))
  rng = get_int_range(t)
  if rng is not None:
    lo, hi = rng
    if hi <= 255 and (lo >= 0 or (lo == -1 and float(void_class) == -1.0)):
      return True
  ent = getattr(t, '_se3ds_byte_range', None)
  if ent is not None and ent[1:4] == same and (ent[4] is None or ent[4] == float(void_class)):
    return ent[0]
  bad = torch.empty(1, dtype=torch.int32, device=t.device)
  tc = t.contiguous()
  rc = _lib.lib().se3ds_feats_byte_range(_lib.ptr(tc), _lib.dtype_code(tc), tc.numel(),
                                         float(void_class), _lib.ptr(bad), _lib.stream())
  _lib.check(rc, 'se3ds_feats_byte_range')
  ok = int(bad.item()) == 0
  set_byte_range(t, ok, void_class)
  return ok


def _splat(entry: str, coords, offset, feats, height, width, depth_scale, input_void_class,
           output_void_class, with_mask=False, mask_void=constants.INVALID_RGB_VALUE):
  if feats.dim() != 2 and feats.dim() != 3:
    raise ValueError('feats should have shape (N, M) or (N, M, C), got'
                     f' {tuple(feats.shape)} instead.')
  _lib.require_cuda(coords, feats, offset)
  is_scalar = feats.dim() == 2
  if is_scalar:
    feats = feats[..., None]
  if feats.dtype not in (torch.float32, torch.int32, torch.uint8):
    feats = feats.to(torch.float32)
  hint = FEAT_BYTE_RANGE if (feats.dtype == torch.int32 and feats.shape[-1] <= 3 and
                             byte_range(feats, input_void_class)) else 0
  feats = feats.contiguous()
  coords = coords.to(torch.float32).contiguous()
  n, four, m = coords.shape
  if four != 4 or feats.shape[0] != n or feats.shape[1] != m:
    raise ValueError(f'coords {tuple(coords.shape)} and feats {tuple(feats.shape)} disagree')
  c = feats.shape[-1]
  dev = coords.device
  depth = torch.empty((n, height, width), dtype=torch.float32, device=dev)
  out = torch.empty((n, height, width, c), dtype=torch.float32, device=dev)
  mask = torch.empty((n, height, width), dtype=torch.float32, device=dev) if with_mask else None
  L = _lib.lib()
  nbytes = L.se3ds_splat_workspace_bytes(n, m, height, width, c)
  ws = _workspace(nbytes, dev)
  if entry == 'equirect':
    if offset is not None:
      offset = offset.to(torch.float32).contiguous()
    rc = L.se3ds_project_equirect(_lib.ptr(coords), _lib.ptr(offset), _lib.ptr(feats),
                                  _lib.dtype_code(feats) | hint, n, m, c, height, width,
                                  float(depth_scale), float(input_void_class),
                                  float(output_void_class), _lib.ptr(depth), _lib.ptr(out),
                                  _lib.ptr(mask), float(mask_void), _lib.ptr(ws), ws.numel(),
                                  _lib.stream())
    _lib.check(rc, 'se3ds_project_equirect')
  else:
    rc = L.se3ds_project_to_feat(_lib.ptr(coords), _lib.ptr(feats), _lib.dtype_code(feats) | hint, n, m,
                                 c, height, width, float(depth_scale), float(input_void_class),
                                 float(output_void_class), _lib.ptr(depth), _lib.ptr(out),
                                 _lib.ptr(mask), float(mask_void), _lib.ptr(ws), ws.numel(),
                                 _lib.stream())
    _lib.check(rc, 'se3ds_project_to_feat')
  if hint:
    _poll_promise(dev)
  if is_scalar:
    out = out[..., 0]
  if with_mask:
    return depth, out, mask
  return depth, out


def project_to_feat(transformed_coords: torch.Tensor, feats: torch.Tensor, height: int,
                    width: int, depth_scale: float, input_void_class: float,
                    output_void_class: float = 0) -> Tuple[torch.Tensor, torch.Tensor]:
  """Z-buffer splat of point features (reference :90-183).

  transformed_coords (N,4,M) (x, y, z, 1); feats (N,M) or (N,M,C).  Returns projected depth
  (N,H,W) in [0,1] and projected feats (N,H,W[,C]) fp32.  Reference quirks are kept: invalid
  and culled points scatter into flat index 0; 0.1 m tolerance; per-channel max."""
  return _splat('generic', transformed_coords, None, feats, height, width, depth_scale,
                input_void_class, output_void_class)


def splat_debug_indices(n: int, m: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
  """(idx, z) of the last splat on `device` (parity tap; idx = v*W+u or -1 for the sink)."""
  ws = _workspaces[_dev_key(device)]
  idx = torch.empty((n, m), dtype=torch.int32, device=device)
  z = torch.empty((n, m), dtype=torch.float32, device=device)
  rc = _lib.lib().se3ds_splat_debug_indices(_lib.ptr(ws), n, m, _lib.ptr(idx), _lib.ptr(z),
                                            _lib.stream())
  _lib.check(rc, 'se3ds_splat_debug_indices')
  return idx, z


def debug_fast_fxy(xyz: torch.Tensor, height: int, width: int):
  """Parity tap: (fx, fy, verdict) of the device's fast index screen on camera-relative xyz (3,M)
  (verdict >= -1: decided flat index or -1; -2: left to the exact binary64 chain)."""
  _lib.require_cuda(xyz)
  xyz = xyz.to(torch.float32).contiguous()
  m = xyz.shape[1]
  fx = torch.empty(m, dtype=torch.float32, device=xyz.device)
  fy = torch.empty_like(fx)
  verdict = torch.empty(m, dtype=torch.int32, device=xyz.device)
  rc = _lib.lib().se3ds_debug_fast_fxy(_lib.ptr(xyz), m, width, height, _lib.ptr(fx), _lib.ptr(fy),
                                       _lib.ptr(verdict), _lib.stream())
  _lib.check(rc, 'se3ds_debug_fast_fxy')
  return fx, fy, verdict


class PointCloudMemory:
  """The growing point-cloud memory of the trajectory loops -- `memory_coords (N,4,M)` /
  `memory_feats (N,M,C)` of utils/eval_metric.py:144-239, trainers/gan_manager.py:462-541 -- kept
  in preallocated HBM buffers: frames are unprojected straight into the next free window
  (`se3ds_unproject_equirect_into`) and rendered from the first M points
  (`se3ds_project_equirect_memory`), so the reference's `tf.concat` per frame (a copy of the whole
  memory, O(T^2) traffic over a trajectory) never happens.  Capacity doubles when exhausted."""

  def __init__(self, batch_size: int, channels: int, dtype=torch.int32, device='cuda:0',
               capacity: int = 0):
    self.n, self.c, self.dtype, self.device = batch_size, channels, dtype, torch.device(_dev_key(device))
    self.m = 0
    self.byte_range = True   # every appended frame's features were void or in [0, 255]
    self._x = torch.empty((batch_size, 4, 0), dtype=torch.float32, device=self.device)
    self._f = torch.empty((batch_size, 0, channels), dtype=dtype, device=self.device)
    if capacity:
      self.reserve(capacity)

  @property
  def capacity(self) -> int:
    return self._x.shape[2]

  @property
  def coords(self) -> torch.Tensor:
    """(N,4,M) view of the valid part."""
    return self._x[:, :, :self.m]

  @property
  def feats(self) -> torch.Tensor:
    return self._f[:, :self.m]

  def reserve(self, capacity: int):
    if capacity <= self.capacity:
      return
    cap = max(capacity, 2 * self.capacity)
    x = torch.empty((self.n, 4, cap), dtype=torch.float32, device=self.device)
    f = torch.empty((self.n, cap, self.c), dtype=self.dtype, device=self.device)
    # the homogeneous row is 1.0 for every point ever written (models.py:225-226 adds 0 to it): filled
    # ONCE here, so that the unprojects write 40 instead of 44 bytes per pixel (SE3DS_XYZ1_ONES_PRESET)
    x[:, 3, :].fill_(1.0)
    if self.m:
      x[:, :, :self.m].copy_(self._x[:, :, :self.m])
      f[:, :self.m].copy_(self._f[:, :self.m])
    self._x, self._f = x, f

  def append_equirect(self, feats: torch.Tensor, depth: torch.Tensor, void_class: float,
                      depth_scale: float, position: Optional[torch.Tensor] = None):
    """equirectangular_to_pointcloud(feats, depth) (+ position) appended in place."""
    from se3ds_amd.utils import pano_utils
    p = depth.shape[1] * depth.shape[2]
    self.reserve(self.m + p)
    if feats.dim() == 3:
      feats = feats[..., None]
    # (the unprojected features are the source's, or void: the promise carries over)
    self.byte_range = self.byte_range and byte_range(feats, void_class)
    pano_utils.equirectangular_to_pointcloud(feats, depth, void_class, depth_scale,
                                             position=position, out=(self._x, self._f, self.m),
                                             ones_preset=True)
    self.m += p

  def clear(self):
    """Forgets the points (the buffers stay)."""
    self.m = 0
    self.byte_range = True

  def append_views_and_project(self, views, void_class: float, depth_scale: float,
                               target: Optional[torch.Tensor], height: int, width: int,
                               with_mask: bool = False,
                               mask_void: float = constants.INVALID_RGB_VALUE,
                               output_void_class: float = 0, out=None):
    """One trajectory step in ONE library call (`se3ds_warp_views_to_target`): every view
    (feats (N,H,W[,C]), depth (N,H,W), position (N,3) or None) is appended as append_equirect does
    and the target at `target` (N,3) is rendered as project() does -- utils/eval_metric.py:153-166,
    233-240; the launches are queued back to back instead of through four Python calls.
    out = (depth (N,H,W), feats (N,H,W,C), mask (N,H,W) or None) fp32: render into these tensors
    instead of new ones (a trajectory loop reuses its frame buffers)."""
    if not views:
      return self.project(height, width, void_class, depth_scale, position=target,
                          with_mask=with_mask, mask_void=mask_void,
                          output_void_class=output_void_class)
    feats0 = views[0][0]
    vh, vw = views[0][1].shape[1], views[0][1].shape[2]
    if void_class < 0.0 and feats0.dtype == torch.uint8:
      raise ValueError('feats datatype must be signed if the void class is negative')
    p = vh * vw
    self.reserve(self.m + len(views) * p)
    fl, dl, pl, keep = [], [], [], []
    br_new = self.byte_range   # (committed with self.m once the call has succeeded: ADVICE r5)
    any_pos = any(v[2] is not None for v in views)
    for feats, depth, pos in views:
      if feats.dim() == 3:
        feats = feats[..., None]
      _lib.require_cuda(feats, depth, pos)
      if (feats.dtype != self.dtype or tuple(feats.shape) != (self.n, vh, vw, self.c) or
          tuple(depth.shape) != (self.n, vh, vw)):
        raise ValueError('views must share the memory\'s dtype, batch, channels and one size')
      br_new = br_new and byte_range(feats, void_class)
      feats, depth = feats.contiguous(), depth.to(torch.float32).contiguous()
      if any_pos:
        pos = (torch.zeros((self.n, 3), dtype=torch.float32, device=self.device) if pos is None
               else pos.to(torch.float32).contiguous())
        pl.append(pos.data_ptr())
      keep += [feats, depth, pos]
      fl.append(feats.data_ptr())
      dl.append(depth.data_ptr())
    import ctypes
    nv = len(views)
    arr = lambda xs: (ctypes.c_void_p * nv)(*xs)
    dev, n, c = self.device, self.n, self.c
    tab = _host_tables.equirect_tables(vh, vw, dev)
    base = tab.data_ptr()
    if out is not None:
      depth_o, out, mask = out
      if not with_mask:
        mask = None
      if with_mask and mask is None:
        raise ValueError('with_mask needs out = (depth, feats, mask)')
      # (raw pointers from here on: every buffer is checked for device, dtype, shape and layout)
      _lib.require_cuda(depth_o, out, mask)
      for t, shape in ((depth_o, (n, height, width)), (out, (n, height, width, c)),
                       (mask, (n, height, width))):
        if t is not None and (tuple(t.shape) != shape or t.dtype != torch.float32 or
                              not t.is_contiguous() or _dev_key(t.device) != _dev_key(dev)):
          raise ValueError('out = (depth (N,H,W), feats (N,H,W,C), mask (N,H,W) or None): fp32, '
                           f'contiguous, on {dev}; got {tuple(t.shape)} {t.dtype} on {t.device}')
    else:
      depth_o = torch.empty((n, height, width), dtype=torch.float32, device=dev)
      out = torch.empty((n, height, width, c), dtype=torch.float32, device=dev)
      mask = torch.empty((n, height, width), dtype=torch.float32, device=dev) if with_mask else None
    L = _lib.lib()
    m_new = self.m + nv * p
    key = (n, m_new, height, width, c)
    nbytes = _WS_BYTES.get(key)
    if nbytes is None:   # (a ctypes call per step otherwise; the answer only depends on the key)
      nbytes = _WS_BYTES[key] = L.se3ds_splat_workspace_bytes(n, m_new, height, width, c)
    ws = _workspace(nbytes, dev)
    if target is not None:
      _lib.require_cuda(target)
      target = target.to(torch.float32).contiguous()
    hint = FEAT_BYTE_RANGE if (self.dtype == torch.int32 and c <= 3 and br_new) else 0
    rc = L.se3ds_warp_views_to_target(
        arr(fl), _lib.dtype_code(self._f) | hint | XYZ1_ONES_PRESET, arr(dl), arr(pl) if any_pos else None,
        nv, n, vh, vw,
        c, float(void_class), float(depth_scale), base, base + 4 * vh, base + 8 * vh,
        base + 8 * vh + 4 * vw, _lib.ptr(self._x), _lib.ptr(self._f), self.capacity, self.m,
        _lib.ptr(target), height, width, float(output_void_class), _lib.ptr(depth_o), _lib.ptr(out),
        _lib.ptr(mask), float(mask_void), _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, 'se3ds_warp_views_to_target')
    self.m, self.byte_range = m_new, br_new
    if hint:
      _poll_promise(dev)
    return (depth_o, out, mask) if with_mask else (depth_o, out)

  def project(self, height: int, width: int, void_class: float, depth_scale: float,
              position: Optional[torch.Tensor] = None, with_mask: bool = False,
              mask_void: float = constants.INVALID_RGB_VALUE, output_void_class: float = 0):
    """project_feats_to_equirectangular(memory_feats, memory_coords - position, ...)."""
    _lib.require_cuda(position)
    dev = self.device
    n, c = self.n, self.c
    depth = torch.empty((n, height, width), dtype=torch.float32, device=dev)
    out = torch.empty((n, height, width, c), dtype=torch.float32, device=dev)
    mask = torch.empty((n, height, width), dtype=torch.float32, device=dev) if with_mask else None
    L = _lib.lib()
    ws = _workspace(L.se3ds_splat_workspace_bytes(n, self.m, height, width, c), dev)
    if position is not None:
      position = position.to(torch.float32).contiguous()
    x, f = self._x, self._f
    if self.capacity == 0:   # empty memory: any valid pointer pair will do (m = 0)
      x = torch.empty((n, 4, 1), dtype=torch.float32, device=dev)
      f = torch.empty((n, 1, c), dtype=self.dtype, device=dev)
    hint = FEAT_BYTE_RANGE if (f.dtype == torch.int32 and c <= 3 and self.byte_range) else 0
    rc = L.se3ds_project_equirect_memory(
        _lib.ptr(x), _lib.ptr(position), _lib.ptr(f), _lib.dtype_code(f) | hint, n, self.m, x.shape[2], c,
        height, width, float(depth_scale), float(void_class), float(output_void_class),
        _lib.ptr(depth), _lib.ptr(out), _lib.ptr(mask), float(mask_void), _lib.ptr(ws), ws.numel(),
        _lib.stream())
    _lib.check(rc, 'se3ds_project_equirect_memory')
    if hint:
      _poll_promise(dev)
    return (depth, out, mask) if with_mask else (depth, out)
