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
