"""Model configuration of the SE3DS inference wrapper.

The drop-in boundary fixes the NAMES, defaults and factory functions of the
reference's models/model_config.py:22-61 -- `SE3DSConfig` with its twelve attributes,
`get_config()`, `get_re10k_config()`, `get_test_config()` -- because callers set and read them by
name (an invalid value surfaces where the reference raises: when the model is built).  Everything
else here is this repository's own: a dataclass and one table of
per-variant overrides instead of three hand-written functions.
"""
import dataclasses
from typing import Optional

from se3ds_amd import constants


@dataclasses.dataclass
class SE3DSConfig:
  """Attributes and defaults as the reference's class of the same name (models/model_config.py:22-36)."""
  batch_size: int = 1                                   # SE3DSModel raises unless 1 (models.py:95-96)
  ckpt_path: Optional[str] = constants.CKPT_UNSEEN      # None: random initialisation
  hidden_dims: int = 128
  random_noise: bool = True
  z_dim: int = 32
  circular_pad: bool = True                             # inference pads W by wrapping (layers.py:67-80)
  depth_scale: float = constants.DEPTH_SCALE            # metres at depth 1.0
  gen_dims: int = 128
  image_height: int = 512                               # panoramas are image_height x 2 image_height
  h_fov: float = 0.17
  resnet_version: str = '101'
  use_blurred_mask: bool = True                         # 5 generator input channels instead of 4


# variant -> overrides of the defaults above (reference :39-61)
_VARIANTS = {
    'val_unseen': dict(ckpt_path=constants.CKPT_UNSEEN, resnet_version='101'),
    're10k': dict(ckpt_path=constants.CKPT_RE10K, resnet_version='101', use_blurred_mask=False),
    'test': dict(ckpt_path=None, hidden_dims=4, z_dim=4, gen_dims=4),
}


def _make(variant: str) -> SE3DSConfig:
  return SE3DSConfig(**_VARIANTS[variant])


def get_config() -> SE3DSConfig:
  """The Val-Unseen (Matterport3D) configuration."""
  return _make('val_unseen')


def get_re10k_config() -> SE3DSConfig:
  """The RealEstate10K configuration (no blurred-mask input channel)."""
  return _make('re10k')


def get_test_config() -> SE3DSConfig:
  """Tiny dimensions, no checkpoint: what the reference's unit tests build."""
  return _make('test')
