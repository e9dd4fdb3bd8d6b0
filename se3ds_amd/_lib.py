"""ctypes binding of libse3ds_hip.so (include/se3ds_hip.h).  There is NO fallback: if the
library is missing or a GPU entry point fails, the caller gets an exception."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SE3DS_LIB: another build of the library for this process (the A/B scripts under tools/probes/ compare
# two builds on one box; they used to copy over the in-tree file and a killed job left the wrong one
# behind -- ADVICE r5).  Unset: the in-tree build, the only one the tests and the bench ever load.
SO_PATH = os.environ.get('SE3DS_LIB') or os.path.join(_HERE, 'csrc', 'libse3ds_hip.so')

F32, I32, U8, BF16 = 0, 1, 2, 3
_DTYPE_CODE = {torch.float32: F32, torch.int32: I32, torch.uint8: U8, torch.bfloat16: BF16}

c_int, c_i64, c_f, c_p, c_sz = (ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p,
                                ctypes.c_size_t)

# name -> (restype, argtypes); mirrors include/se3ds_hip.h one to one.
_SIGS = {
    'se3ds_version': (ctypes.c_char_p, []),
    'se3ds_last_error': (ctypes.c_char_p, []),
    'se3ds_unproject_equirect': (c_int, [c_p, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int,
                                         c_int, c_int, c_f, c_f, c_p, c_p, c_p]),
    'se3ds_unproject_equirect_into': (c_int, [c_p, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int,
                                              c_int, c_int, c_f, c_f, c_p, c_p, c_i64, c_i64, c_p]),
    'se3ds_splat_workspace_bytes': (c_sz, [c_int, c_i64, c_int, c_int, c_int]),
    'se3ds_project_equirect': (c_int, [c_p, c_p, c_p, c_int, c_int, c_i64, c_int, c_int, c_int,
                                       c_f, c_f, c_f, c_p, c_p, c_p, c_f, c_p, c_sz, c_p]),
    'se3ds_project_equirect_memory': (c_int, [c_p, c_p, c_p, c_int, c_int, c_i64, c_i64, c_int, c_int,
                                              c_int, c_f, c_f, c_f, c_p, c_p, c_p, c_f, c_p, c_sz, c_p]),
    'se3ds_warp_views_to_target': (c_int, [c_p, c_int, c_p, c_p, c_int, c_int, c_int, c_int, c_int,
                                           c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_p,
                                           c_int, c_int, c_f, c_p, c_p, c_p, c_f, c_p, c_sz, c_p]),
    'se3ds_project_to_feat': (c_int, [c_p, c_p, c_int, c_int, c_i64, c_int, c_int, c_int, c_f, c_f,
                                      c_f, c_p, c_p, c_p, c_f, c_p, c_sz, c_p]),
    'se3ds_splat_debug_indices': (c_int, [c_p, c_int, c_i64, c_p, c_p, c_p]),
    'se3ds_splat_promise_sticky': (c_int, [c_p, c_p, c_int, c_p]),
    'se3ds_feats_byte_range': (c_int, [c_p, c_int, c_i64, c_f, c_p, c_p]),
    'se3ds_splat_promise_broken': (c_int, [c_p, c_int, c_i64, c_p, c_p]),
    'se3ds_debug_fast_fxy': (c_int, [c_p, c_i64, c_int, c_int, c_p, c_p, c_p, c_p]),
    'se3ds_unproject_perspective': (c_int, [c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int,
                                            c_f, c_p, c_p, c_p]),
    'se3ds_interp_bilinear': (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_i64, c_int, c_p,
                                      c_p]),
    'se3ds_rotate_coords': (c_int, [c_p, c_p, c_int, c_i64, c_int, c_int, c_p, c_p]),
    'se3ds_perspective_coords': (c_int, [c_p, c_p, c_i64, c_int, c_f, c_p, c_p]),
    'se3ds_persp_from_equirect_coords': (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_p, c_p]),
    'se3ds_input_transform': (c_int, [c_p] * 9 + [c_int] * 5 + [c_p] * 8),
    'se3ds_perspective_to_pointcloud': (c_int, [c_p, c_p, c_int, c_int, c_int, c_p, c_p, c_int, c_f, c_p, c_p,
                                                c_p, c_p, c_p, c_int, c_int, c_f, c_f, c_p, c_p, c_p]),
    'se3ds_perspective_guidance': (c_int, [c_p, c_p, c_int, c_int, c_p, c_p, c_int, c_int, c_p, c_p, c_p,
                                           c_p]),
    'se3ds_mask_pano': (c_int, [c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_f, c_p, c_p]),
    'se3ds_resize': (c_int, [c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p, c_p]),
    'se3ds_mean_f32': (c_int, [c_p, c_i64, c_p, c_p]),
    'se3ds_compact_workspace_bytes': (c_sz, [c_i64]),
    'se3ds_compact_valid': (c_int, [c_p, c_p, c_int, c_int, c_i64, c_int, c_f, c_p, c_p, c_p, c_p,
                                    c_sz, c_p]),
}

_lib = None


class Se3dsHipError(RuntimeError):
  pass


def declared_symbols():
  return sorted(_SIGS)


def register(sigs):
  """Lets other binding modules (conv, norm, ...) add their signatures before load."""
  _SIGS.update(sigs)
  if _lib is not None:
    _bind(_lib, sigs)


def _bind(lib, sigs):
  for name, (res, args) in sigs.items():
    fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
    fn.restype = res
    fn.argtypes = args


def lib():
  global _lib
  if _lib is None:
    if not os.path.exists(SO_PATH):
      raise Se3dsHipError(
          f'{SO_PATH} not found: build it with `python -m se3ds_amd.csrc.build` '
          '(__graft_entry__.build()).  There is no CPU fallback for the SE3DS hot path.')
    l = ctypes.CDLL(SO_PATH)
    _bind(l, _SIGS)
    _lib = l
  return _lib


def dtype_code(t):
  try:
    return _DTYPE_CODE[t.dtype]
  except KeyError:
    raise ValueError(f'unsupported dtype {t.dtype}')


def ptr(t):
  return None if t is None else t.data_ptr()


def stream():
  return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
  for t in tensors:
    if t is not None and not t.is_cuda:
      raise Se3dsHipError('SE3DS hot-path ops need tensors on an MI355X (cuda) device; got a '
                          f'{t.device} tensor.  There is no CPU fallback.')


def check(rc, what):
  if rc != 0:
    msg = lib().se3ds_last_error().decode()
    names = {-1: 'BADSHAPE', -2: 'BADDTYPE', -3: 'WORKSPACE', -4: 'LAUNCH', -5: 'UNSUPPORTED'}
    raise Se3dsHipError(f'{what} failed: SE3DS_E_{names.get(rc, rc)} {msg}')
