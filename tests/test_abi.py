"""CPU test: libse3ds_hip.so builds for gfx950, loads, and exports every symbol that
include/se3ds_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

from se3ds_amd import _lib
from se3ds_amd.csrc import build as hip_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
  text = open(os.path.join(ROOT, 'include', 'se3ds_hip.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(se3ds_[a-z0-9_]+)\s*\(', text)))


def test_library_builds_and_exports_every_declared_symbol():
  so = hip_build.build()
  assert os.path.exists(so)
  lib = ctypes.CDLL(so)
  names = _declared()
  assert len(names) >= 10
  for name in names:
    assert hasattr(lib, name), f'{name} declared in include/se3ds_hip.h but not exported'
  assert lib.se3ds_version
  lib.se3ds_version.restype = ctypes.c_char_p
  assert b'gfx950' in lib.se3ds_version()


def test_python_binding_table_matches_header():
  import se3ds_amd.hipops  # noqa: F401  registers the conv / norm / optimiser signatures
  assert sorted(_lib.declared_symbols()) == _declared()
  _lib.lib()  # binds every signature; AttributeError if one is missing


def test_no_cpu_fallback():
  import pytest
  import torch
  from se3ds_amd.utils import pano_utils
  with pytest.raises(_lib.Se3dsHipError):
    pano_utils.mask_pano(torch.zeros((1, 8, 16, 3)))


def test_fastdiv_matches_integer_division():
  """The conv kernels turn a tile row into (image, row, column) with host-made multipliers instead of
  divisions (csrc/conv.hip FastDiv: q = (t + ((n - t) >> s1)) >> s2, t = mulhi(n, m)).  The same
  arithmetic evaluated on the host (se3ds_fastdiv_host) must equal n // d for every divisor class:
  1, powers of two, odd, the image sizes of the configs, large -- at the edges of the 32-bit range."""
  import random
  import se3ds_amd.hipops  # noqa: F401
  L = _lib.lib()
  rng = random.Random(5)
  divisors = [1, 2, 3, 5, 7, 16, 17, 31, 32, 33, 64, 129, 255, 256, 257, 512, 513, 1024, 2048, 32 * 64,
              129 * 257, 512 * 1024, 1024 * 2048, 1000003, 2 ** 20 + 1, 2 ** 30, 2 ** 31 - 1, 2 ** 31,
              2 ** 32 - 1] + [rng.randrange(1, 2 ** 31) for _ in range(40)]
  for d in divisors:
    ns = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1]
    ns += [k * d + r for k in (3, 1000, 65535) for r in (-1, 0, 1)]
    ns += [rng.randrange(0, 2 ** 32) for _ in range(200)]
    for n in ns:
      if 0 <= n < 2 ** 32:
        assert L.se3ds_fastdiv_host(n, d) == n // d, (n, d)
